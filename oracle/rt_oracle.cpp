// rt_oracle.cpp — CPU restatement of the reference path tracer.  TEST INFRASTRUCTURE ONLY
// (see rt_oracle.h).  PARITY UNPINNED BY THE REFERENCE: it ships no tests or goldens for this
// path and cannot be built here; known answers in tests/golden/ are hand-derived.
//
// Structure follows the reference on purpose — virtual Hittable / Material / Texture / Pdf
// objects, recursive ray_color, HittableList scans, the reference's own BvhNode — so that it
// reads against /root/reference/raytracer/src file by file.  All arithmetic is f32 with the
// reference's operation order; build with -ffp-contract=off.  Citations are
// `file:line` relative to /root/reference/raytracer/src unless they start with math/.
//
// Two things the reference does not have are taken from include/: the seed schedule
// (rt_seed_schedule.h: which random number each call site gets) and the elementary functions
// (rt_portable_math.h; `use_libm` switches back to glibc for comparison).
#include "rt_oracle.h"

#include <atomic>
#include <cmath>
#include <limits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>

#include "rt_portable_math.h"
#include "rt_seed_schedule.h"

namespace {

thread_local std::string g_error;
void set_error(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}

typedef float Real;

/* ------------------------------------------------------------------ math/src/vec3.rs */
struct Vec3 {
    Real x, y, z;
    Real operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
};
inline Vec3 V(Real x, Real y, Real z) { return Vec3{x, y, z}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, Real k) { return {a.x * k, a.y * k, a.z * k}; }
inline Vec3 operator*(Real k, Vec3 a) { return a * k; }  // vec3.rs:553-555: rhs * self
inline Vec3 operator/(Vec3 a, Real k) { return {a.x / k, a.y / k, a.z / k}; }
inline Real length_squared(Vec3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }  // vec3.rs:638-643
inline Real length(Vec3 v) { return std::sqrt(length_squared(v)); }               // vec3.rs:645-655
inline Real dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }     // vec3.rs:711-716
inline Vec3 cross(Vec3 a, Vec3 b) {                                                // vec3.rs:728-737
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline Vec3 normalize(Vec3 v) {  // vec3.rs:690-700
    Real lensq = length_squared(v);
    if (lensq == 0.0f) return V(0, 0, 0);
    return v * (1.0f / std::sqrt(lensq));
}
inline Vec3 reflect_unit_vector(Vec3 v, Vec3 n) {  // vec3.rs:740-753
    Real two = 1.0f + 1.0f;
    return v - two * dot(v, n) * n;
}
inline Vec3 refract(Vec3 uv, Vec3 n, Real etai_over_etat) {  // vec3.rs:756-771 (no abs under the root)
    Real cos_theta = dot(-uv, n);
    Vec3 r_out_parallel = etai_over_etat * (uv + cos_theta * n);
    Vec3 r_out_perp = -std::sqrt(1.0f - length_squared(r_out_parallel)) * n;
    return r_out_parallel + r_out_perp;
}
inline bool are_on_the_same_plane_side(Vec3 a, Vec3 b) { return dot(a, b) > 0.0f; }  // vec3.rs:789-801

struct Color {  // math/src/colors.rs (RGB only: alpha is a meaningless running sum)
    Real r, g, b;
};
inline Color C(Real r, Real g, Real b) { return Color{r, g, b}; }
inline Color operator+(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Color operator*(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Color operator*(Color a, Real k) { return {a.r * k, a.g * k, a.b * k}; }
inline Color operator/(Color a, Real k) { return {a.r / k, a.g / k, a.b / k}; }

struct Ray {  // math/src/ray.rs:10-40
    Vec3 origin, direction;
    Real time;
    Vec3 at(Real t) const { return origin + t * direction; }
};

/* ------------------------------------------------------------------ types.rs */
thread_local rts::Rng *tl_rng = nullptr;
thread_local uint64_t tl_rays = 0;
bool g_use_libm = false;

inline Real m_sin(Real x) { return g_use_libm ? sinf(x) : pm::sinf_(x); }
inline Real m_cos(Real x) { return g_use_libm ? cosf(x) : pm::cosf_(x); }
inline Real m_acos(Real x) { return g_use_libm ? acosf(x) : pm::acosf_(x); }
inline Real m_atan2(Real y, Real x) { return g_use_libm ? atan2f(y, x) : pm::atan2f_(y, x); }
inline Real m_ln(Real x) { return g_use_libm ? logf(x) : pm::logf_(x); }
inline void m_sincos(Real x, Real *s, Real *c) {
    if (g_use_libm) { *s = sinf(x); *c = cosf(x); } else pm::sincosf_(x, s, c);
}

const Real C_PI = 3.14159265358979323846f;
inline Real degrees_to_radians(Real d) { return (d * C_PI) / 180.0f; }       // types.rs:44-46
// Known-answer hook (orc_bounce): a caller-supplied sequence of "random" numbers instead of the stream
thread_local const float *tl_forced = nullptr;
thread_local int tl_forced_left = 0, tl_forced_used = 0;
inline Real random_real() {                                                     // types.rs:48-50
    if (tl_forced) {
        ++tl_forced_used;
        if (tl_forced_left > 0) { --tl_forced_left; return *tl_forced++; }
        return 0.0f;                                                            // (ran out: the caller sees used > given)
    }
    return rts::random_real(*tl_rng);
}
inline Real random_real_range(Real lo, Real hi) {                              // :52-55: rand's UniformFloat = u * (hi - lo) + lo
    if (tl_forced) { Real u = random_real(), scale = hi - lo, v = u * scale; return v + lo; }
    return rts::random_real_range(*tl_rng, lo, hi);
}
inline int random_int(int lo, int hi) {                                                               // :106-108
    if (tl_forced) { int k = lo + (int)(random_real() * (Real)(hi - lo + 1)); return k > hi ? hi : k; }   // (hook: xi picks the object)
    return rts::random_int(*tl_rng, lo, hi);
}
inline Vec3 rand_vec3_range(Real lo, Real hi) {                                 // types.rs:87-93
    Real x = random_real_range(lo, hi), y = random_real_range(lo, hi), z = random_real_range(lo, hi);
    return V(x, y, z);
}
inline Vec3 random_in_unit_sphere() {  // types.rs:95-104
    for (;;) {
        Vec3 p = rand_vec3_range(-1.0f, 1.0f);
        if (length_squared(p) >= 1.0f) continue;
        return p;
    }
}
inline Vec3 random_cosine_direction() {  // types.rs:141-151
    Real r1 = random_real(), r2 = random_real();
    Real z = std::sqrt(1.0f - r2);
    Real phi = (Real)(2.0 * 3.14159265358979323846) * r1;
    Real x = m_cos(phi) * std::sqrt(r2), y = m_sin(phi) * std::sqrt(r2);
    return V(x, y, z);
}
inline Vec3 random_to_sphere(Real radius, Real distance_squared) {  // types.rs:153-163
    Real r1 = random_real(), r2 = random_real();
    Real z = 1.0f + r2 * (std::sqrt(1.0f - radius * radius / distance_squared) - 1.0f);
    Real phi = (Real)(2.0 * 3.14159265358979323846) * r1;
    Real x = m_cos(phi) * std::sqrt(1.0f - z * z), y = m_sin(phi) * std::sqrt(1.0f - z * z);
    return V(x, y, z);
}
inline Real powi5(Real a) {  // f32::powi(5) = llvm.powi: square-and-multiply
    Real a2 = a * a, a4 = a2 * a2;
    return a * a4;
}
inline Real schlick(Real cosine, Real refraction_index) {  // types.rs:165-169
    Real r0 = (1.0f - refraction_index) / (1.0f + refraction_index);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powi5(1.0f - cosine);
}
inline Real ffmin(Real a, Real b) { return a < b ? a : b; }  // types.rs:171-177
inline Real ffmax(Real a, Real b) { return a > b ? a : b; }  // types.rs:179-185
inline Real f32_min(Real a, Real b) { return a != a ? b : (b != b ? a : (a < b ? a : b)); }  // f32::min
inline Real f32_max(Real a, Real b) { return a != a ? b : (b != b ? a : (a > b ? a : b)); }  // f32::max
inline Real f32_clamp(Real x, Real lo, Real hi) { return x < lo ? lo : (x > hi ? hi : x); }  // f32::clamp
const Real F32_MAX = 3.40282347e+38f;

/* ------------------------------------------------------------------ aabb3.rs */
struct Aabb {
    Vec3 min, max;
    bool hit(const Ray &r, Real tmin, Real tmax) const {  // aabb3.rs:19-45
        for (int a = 0; a < 3; ++a) {
            Real inv_d = 1.0f / r.direction[a];
            Real t0 = ffmin((min[a] - r.origin[a]) * inv_d, (max[a] - r.origin[a]) * inv_d);
            Real t1 = ffmax((min[a] - r.origin[a]) * inv_d, (max[a] - r.origin[a]) * inv_d);
            tmin = ffmax(t0, tmin);
            tmax = ffmin(t1, tmax);
            if (tmax <= tmin) return false;
        }
        return true;
    }
    void add_point(Vec3 p) {  // aabb3.rs:47-50 (math::vec3::min / max_sv: component-wise)
        min = V(f32_min(min.x, p.x), f32_min(min.y, p.y), f32_min(min.z, p.z));
        max = V(f32_max(max.x, p.x), f32_max(max.y, p.y), f32_max(max.z, p.z));
    }
};
inline Aabb aabb_default() { return Aabb{V(F32_MAX, F32_MAX, F32_MAX), V(-F32_MAX, -F32_MAX, -F32_MAX)}; }  // :72-79
inline Aabb merge_aabbs(const Aabb &a, const Aabb &b) {  // aabb3.rs:56-70
    return Aabb{V(f32_min(a.min.x, b.min.x), f32_min(a.min.y, b.min.y), f32_min(a.min.z, b.min.z)),
                V(f32_max(a.max.x, b.max.x), f32_max(a.max.y, b.max.y), f32_max(a.max.z, b.max.z))};
}

/* ------------------------------------------------------------------ hittable.rs / material.rs / texture.rs / pdf.rs */
struct Material;
struct HitRecord {
    Vec3 p, normal;
    Real t;
    const Material *mtl;
    bool front_face;
    Real u, v;
    uint32_t prim_id;  // not in the reference: the numbering of include/rt_scene_graph.h
};
inline HitRecord hit_record_new(Vec3 p, Vec3 outward_normal, const Ray &ray, Real t, const Material *mtl, Real u, Real v,
                                uint32_t prim_id) {  // hittable.rs:17-41
    bool front_face = dot(ray.direction, outward_normal) < 0.0f;
    return HitRecord{p, front_face ? outward_normal : -outward_normal, t, mtl, front_face, u, v, prim_id};
}

struct Hittable {
    virtual ~Hittable() {}
    virtual bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const = 0;
    virtual bool bounding_box(Real time0, Real time1, Aabb *out) const = 0;
    virtual Real pdf_value(Vec3, Vec3) const { return 0.0f; }        // hittable.rs:47-49
    virtual Vec3 random(Vec3) const { return V(1.0f, 0.0f, 0.0f); }  // hittable.rs:51-53
    virtual bool is_medium() const { return false; }
};

struct Texture {
    virtual ~Texture() {}
    virtual Color value(Real u, Real v, Vec3 p) const = 0;
};

struct Pdf {
    virtual ~Pdf() {}
    virtual Real value(Vec3 direction) const = 0;
    virtual Vec3 generate() const = 0;
};

struct Onb {  // onb.rs
    Vec3 axis[3];
    Vec3 u() const { return axis[0]; }
    Vec3 v() const { return axis[1]; }
    Vec3 w() const { return axis[2]; }
    Vec3 local_from_vec(Vec3 a) const { return a.x * u() + a.y * v() + a.z * w(); }  // onb.rs:27-29
};
inline Onb onb_from(Vec3 n) {  // onb.rs:32-47
    Vec3 axis_2 = normalize(n);
    Vec3 a = std::fabs(axis_2.x) > 0.9f ? V(0, 1, 0) : V(1, 0, 0);
    Vec3 axis_1 = normalize(cross(axis_2, a));
    Vec3 axis_0 = cross(axis_2, axis_1);
    return Onb{{axis_0, axis_1, axis_2}};
}

struct CosinePdf : Pdf {  // pdf.rs:14-40
    Onb uvw;
    explicit CosinePdf(Vec3 w) : uvw(onb_from(w)) {}
    Real value(Vec3 direction) const override {
        Real cosine = dot(normalize(direction), uvw.w());
        return cosine <= 0.0f ? 0.0f : cosine * (Real)0.318309886183790671537767526745028724;
    }
    Vec3 generate() const override { return uvw.local_from_vec(random_cosine_direction()); }
};

struct HittablePdf : Pdf {  // pdf.rs:42-55
    Vec3 origin;
    const Hittable *obj;
    HittablePdf(const Hittable *o, Vec3 org) : origin(org), obj(o) {}
    Real value(Vec3 direction) const override { return obj->pdf_value(origin, direction); }
    Vec3 generate() const override { return obj->random(origin); }
};

struct MixturePdf : Pdf {  // pdf.rs:57-79
    const Pdf *pdfs[2];
    MixturePdf(const Pdf *p0, const Pdf *p1) : pdfs{p0, p1} {}
    Real value(Vec3 direction) const override { return 0.5f * pdfs[0]->value(direction) + 0.5f * pdfs[1]->value(direction); }
    Vec3 generate() const override { return random_real() < 0.5f ? pdfs[0]->generate() : pdfs[1]->generate(); }
};

struct ScatterRecord {  // material.rs:7-16
    bool is_specular;
    Ray ray;                       // SpecularRec
    std::unique_ptr<Pdf> pdf;      // PdfRec
    Color attenuation;
};

struct Material {
    virtual ~Material() {}
    virtual bool scatter(const Ray &ray, const HitRecord &rec, ScatterRecord *out) const = 0;
    virtual Real scattering_pdf(const Ray &, const HitRecord &, const Ray &) const { return 0.0f; }  // material.rs:21-23
    virtual Color emitted(const Ray &, const HitRecord &, Real, Real, Vec3) const { return C(0, 0, 0); }  // :25-34
};

/* ------------------------------------------------------------------ textures */
struct SolidColorTexture : Texture {  // solid_color_texture.rs
    Color color;
    explicit SolidColorTexture(Color c) : color(c) {}
    Color value(Real, Real, Vec3) const override { return color; }
};

struct CheckerTexture : Texture {  // checker_texture.rs:36-46
    const Texture *odd, *even;
    Real repeat_factor;
    Color value(Real u, Real v, Vec3 p) const override {
        Real sines = m_sin(repeat_factor * p.x) * m_sin(repeat_factor * p.y) * m_sin(repeat_factor * p.z);
        return sines < 0.0f ? odd->value(u, v, p) : even->value(u, v, p);
    }
};

struct PerlinNoise {  // perlin.rs
    const rtg_perlin *tab;
    static Real smooth(Real t) { return t * t * (3.0f - 2.0f * t); }  // perlin.rs:60-62
    static Real perlin_interp(const Vec3 c[2][2][2], Real u, Real v, Real w) {  // perlin.rs:38-58
        Real uu = smooth(u), vv = smooth(v), ww = smooth(w);
        Real accum = 0.0f;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    Vec3 weight = V(u - (Real)i, v - (Real)j, w - (Real)k);
                    accum += ((Real)i * uu + (1.0f - (Real)i) * (1.0f - uu)) * ((Real)j * vv + (1.0f - (Real)j) * (1.0f - vv)) *
                             ((Real)k * ww + (1.0f - (Real)k) * (1.0f - ww)) * dot(c[i][j][k], weight);
                }
        return accum;
    }
    Real noise(Vec3 p) const {  // perlin.rs:84-108 (u,v,w are smoothed here AND in perlin_interp)
        Real u = smooth(p.x - std::floor(p.x)), v = smooth(p.y - std::floor(p.y)), w = smooth(p.z - std::floor(p.z));
        int i = pm::f2i_sat(std::floor(p.x)), j = pm::f2i_sat(std::floor(p.y)), k = pm::f2i_sat(std::floor(p.z));
        Vec3 c[2][2][2];
        for (int di = 0; di < 2; ++di)
            for (int dj = 0; dj < 2; ++dj)
                for (int dk = 0; dk < 2; ++dk) {
                    // i + di wraps in release builds; & 255 keeps the low bits either way
                    int idx = tab->perm_x[(uint32_t)(i + di) & 255u] ^ tab->perm_y[(uint32_t)(j + dj) & 255u] ^
                              tab->perm_z[(uint32_t)(k + dk) & 255u];
                    c[di][dj][dk] = V(tab->ranvec[idx][0], tab->ranvec[idx][1], tab->ranvec[idx][2]);
                }
        return perlin_interp(c, u, v, w);
    }
    Real turbulence(Vec3 p, int depth) const {  // perlin.rs:110-123
        Vec3 temp_p = p;
        Real weight = 1.0f, acc = 0.0f;
        for (int d = 0; d < depth; ++d) {
            acc = acc + weight * noise(temp_p);
            weight *= 0.5f;
            temp_p = temp_p * 2.0f;
        }
        return std::fabs(acc);
    }
};

struct NoiseTexture : Texture {  // noise_texture.rs:21-31
    PerlinNoise perlin;
    Real scale;
    Color value(Real, Real, Vec3 point) const override {
        Real k = 1.0f + m_sin(scale * point.z + 10.0f * perlin.turbulence(point, 7));
        return (C(1, 1, 1) * 0.5f) * k;
    }
};

struct ImageTexture : Texture {  // image_texture.rs:50-86
    uint32_t width = 0, height = 0, bytes_per_scanline = 0;
    const uint8_t *pixels = nullptr;
    Color value(Real u, Real v, Vec3) const override {
        if (!pixels) return C(0.0f, 1.0f, 1.0f);
        u = f32_clamp(u, 0.0f, 1.0f);
        v = 1.0f - f32_clamp(v, 0.0f, 1.0f);
        int i = pm::f2i_sat(u * (Real)width), j = pm::f2i_sat(v * (Real)height);
        if (i >= (int)width) i = (int)width - 1;
        if (j >= (int)height) j = (int)height - 1;
        Real color_scale = 1.0f / 255.0f;
        size_t start = (size_t)j * bytes_per_scanline + (size_t)i * 4;
        return C(color_scale * (Real)pixels[start], color_scale * (Real)pixels[start + 1], color_scale * (Real)pixels[start + 2]);
    }
};

/* ------------------------------------------------------------------ materials */
struct Lambertian : Material {  // lambertian.rs:32-52
    const Texture *albedo;
    bool scatter(const Ray &, const HitRecord &rec, ScatterRecord *out) const override {
        out->is_specular = false;
        out->attenuation = albedo->value(rec.u, rec.v, rec.p);
        out->pdf.reset(new CosinePdf(rec.normal));
        return true;
    }
    Real scattering_pdf(const Ray &, const HitRecord &rec, const Ray &scattered) const override {
        Real cosine = dot(rec.normal, normalize(scattered.direction));
        return cosine < 0.0f ? 0.0f : cosine / C_PI;
    }
};

struct Metal : Material {  // metal.rs:20-37
    Color albedo;
    Real fuzziness;
    bool scatter(const Ray &ray, const HitRecord &rec, ScatterRecord *out) const override {
        Vec3 reflected = reflect_unit_vector(ray.direction, normalize(rec.normal));
        Vec3 scattered = reflected + fuzziness * random_in_unit_sphere();
        if (!are_on_the_same_plane_side(scattered, rec.normal)) return false;
        out->is_specular = true;
        out->ray = Ray{rec.p, scattered, ray.time};
        out->attenuation = albedo;
        return true;
    }
};

struct Dielectric : Material {  // dielectric.rs:16-68
    Real refraction_index;
    bool scatter(const Ray &ray, const HitRecord &rec, ScatterRecord *out) const override {
        Real etai_over_etat = rec.front_face ? 1.0f / refraction_index : refraction_index;
        Vec3 uv = normalize(ray.direction);
        Real cos_theta = f32_min(dot(-uv, rec.normal), 1.0f);
        Real sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
        out->is_specular = true;
        out->attenuation = C(1, 1, 1);
        if (etai_over_etat * sin_theta > 1.0f) {
            out->ray = Ray{rec.p, reflect_unit_vector(uv, normalize(rec.normal)), ray.time};
            return true;
        }
        Real reflect_probability = schlick(cos_theta, etai_over_etat);
        if (random_real() < reflect_probability)
            out->ray = Ray{rec.p, reflect_unit_vector(uv, rec.normal), ray.time};
        else
            out->ray = Ray{rec.p, refract(uv, rec.normal, etai_over_etat), ray.time};
        return true;
    }
};

struct DiffuseLight : Material {  // diffuse_light.rs:34-56
    const Texture *emit;
    bool scatter(const Ray &, const HitRecord &, ScatterRecord *) const override { return false; }
    Color emitted(const Ray &, const HitRecord &rec, Real u, Real v, Vec3 point) const override {
        return rec.front_face ? emit->value(u, v, point) : C(0, 0, 0);
    }
};

struct Isotropic : Material {  // isotropic.rs:25-35 (sic: the new ray's time is hit_record.t)
    const Texture *albedo;
    bool scatter(const Ray &, const HitRecord &rec, ScatterRecord *out) const override {
        out->is_specular = true;
        Vec3 d = random_in_unit_sphere();
        out->ray = Ray{rec.p, d, rec.t};
        out->attenuation = albedo->value(rec.u, rec.v, rec.p);
        return true;
    }
};

/* ------------------------------------------------------------------ objects/sphere.rs */
inline void sphere_get_uv(Vec3 p, Real *u, Real *v) {  // sphere.rs:25-41
    Real theta = m_acos(-p.y);
    Real phi = m_atan2(-p.z, p.x) + (Real)3.14159265358979323846;
    *u = phi / (Real)(2.0 * 3.14159265358979323846);
    *v = theta / (Real)3.14159265358979323846;
}

struct Sphere : Hittable {
    Vec3 center;
    Real radius;
    const Material *mtl;
    uint32_t prim_id;
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override {  // sphere.rs:45-82
        Vec3 oc = r.origin - center;
        Real a = length_squared(r.direction);
        Real half_b = dot(oc, r.direction);
        Real c = length_squared(oc) - radius * radius;
        Real discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0f) return false;
        Real sqrtd = std::sqrt(discriminant);
        Real root = (-half_b - sqrtd) / a;
        if (root < t_min || t_max < root) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || t_max < root) return false;
        }
        Real t = root;
        Vec3 p = r.at(t);
        Vec3 outward_normal = (p - center) / radius;
        Real u, v;
        sphere_get_uv(outward_normal, &u, &v);
        *rec = hit_record_new(p, outward_normal, r, t, mtl, u, v, prim_id);
        return true;
    }
    bool bounding_box(Real, Real, Aabb *out) const override {  // sphere.rs:84-89
        *out = Aabb{center - V(radius, radius, radius), center + V(radius, radius, radius)};
        return true;
    }
    Real pdf_value(Vec3 o, Vec3 v) const override {  // sphere.rs:91-105
        HitRecord rec;
        if (!hit(Ray{o, v, 0.0f}, 0.001f, F32_MAX, &rec)) return 0.0f;
        Real cos_theta_max = std::sqrt(1.0f - radius * radius / length_squared(center - o));
        Real solid_angle = 2.0f * C_PI * (1.0f - cos_theta_max);
        return 1.0f / solid_angle;
    }
    Vec3 random(Vec3 v) const override {  // sphere.rs:107-112
        Vec3 direction = center - v;
        Real distance_squared = length_squared(direction);
        Onb uvw = onb_from(direction);
        return uvw.local_from_vec(random_to_sphere(radius, distance_squared));
    }
};

struct MovingSphere : Hittable {  // sphere.rs:115-227
    Vec3 center0, center1;
    Real radius, time0, time1;
    const Material *mtl;
    uint32_t prim_id;
    Vec3 center(Real time) const { return center0 + ((time - time0) / (time1 - time0)) * (center1 - center0); }  // :144-147
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override {  // :151-188
        Vec3 oc = r.origin - center(r.time);
        Real a = length_squared(r.direction);
        Real half_b = dot(oc, r.direction);
        Real c = length_squared(oc) - radius * radius;
        Real discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0f) return false;
        Real sqrtd = std::sqrt(discriminant);
        Real root = (-half_b - sqrtd) / a;
        if (root < t_min || t_max < root) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || t_max < root) return false;
        }
        Real t = root;
        Vec3 p = r.at(t);
        Vec3 outward_normal = (p - center(r.time)) / radius;
        Real u, v;
        sphere_get_uv(outward_normal, &u, &v);
        *rec = hit_record_new(p, outward_normal, r, t, mtl, u, v, prim_id);
        return true;
    }
    bool bounding_box(Real t0, Real t1, Aabb *out) const override {  // :190-202
        Vec3 rr = V(radius, radius, radius);
        Aabb b0{center(t0) - rr, center(t0) + rr}, b1{center(t1) - rr, center(t1) + rr};
        *out = merge_aabbs(b0, b1);
        return true;
    }
    Real pdf_value(Vec3 o, Vec3 v) const override {  // :204-219
        HitRecord rec;
        if (!hit(Ray{o, v, 0.0f}, 0.001f, F32_MAX, &rec)) return 0.0f;
        Real cos_theta_max = std::sqrt(1.0f - radius * radius / length_squared(center(0.0f) - o));
        Real solid_angle = 2.0f * C_PI * (1.0f - cos_theta_max);
        return 1.0f / solid_angle;
    }
    Vec3 random(Vec3 v) const override {  // :221-226 (sic)
        Vec3 direction = center(time1) - center(time0) - v;
        Real distance_squared = length_squared(direction);
        Onb uvw = onb_from(direction);
        return uvw.local_from_vec(random_to_sphere(radius, distance_squared));
    }
};

/* ------------------------------------------------------------------ rectangles.rs */
// AX = the axis the rectangle is perpendicular to; A, B = the in-plane axes in field order.
template <int A, int B, int AX>
struct Rect : Hittable {
    Real a0, a1, b0, b1, k;
    const Material *mtl;
    uint32_t prim_id;
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override {  // :22-45, 92-115, 162-185
        Real t = (k - r.origin[AX]) / r.direction[AX];
        if (t < t_min || t > t_max) return false;
        Real a = r.origin[A] + t * r.direction[A];
        Real b = r.origin[B] + t * r.direction[B];
        if (a < a0 || a > a1 || b < b0 || b > b1) return false;
        Vec3 n = V(AX == 0 ? 1.0f : 0.0f, AX == 1 ? 1.0f : 0.0f, AX == 2 ? 1.0f : 0.0f);
        *rec = hit_record_new(r.at(t), n, r, t, mtl, (a - a0) / (a1 - a0), (b - b0) / (b1 - b0), prim_id);
        return true;
    }
    bool bounding_box(Real, Real, Aabb *out) const override {  // :47-52, 117-122, 187-192
        Real lo[3], hi[3];
        lo[A] = a0; hi[A] = a1; lo[B] = b0; hi[B] = b1;
        lo[AX] = k - 0.0001f; hi[AX] = k + 0.0001f;
        *out = Aabb{V(lo[0], lo[1], lo[2]), V(hi[0], hi[1], hi[2])};
        return true;
    }
    Real pdf_value(Vec3 origin, Vec3 v) const override {  // :54-67, 124-137, 194-207
        HitRecord rec;
        if (!hit(Ray{origin, v, 0.0f}, 0.001f, F32_MAX, &rec)) return 0.0f;
        Real area = (a1 - a0) * (b1 - b0);
        Real distance_squared = rec.t * rec.t * length_squared(v);
        Real cosine = std::fabs(dot(v, rec.normal) / length(v));
        return distance_squared / (cosine * area);
    }
    Vec3 random(Vec3 origin) const override {  // :69-77, 139-147, 209-217 (fields drawn in x,y,z order)
        Real p[3];
        p[A] = random_real_range(a0, a1);
        p[B] = random_real_range(b0, b1);
        p[AX] = k;
        return V(p[0], p[1], p[2]) - origin;
    }
};
typedef Rect<0, 1, 2> XYRect;
typedef Rect<0, 2, 1> XZRect;
typedef Rect<1, 2, 0> YZRect;

/* ------------------------------------------------------------------ hittable_list.rs */
struct HittableList : Hittable {
    std::vector<const Hittable *> objects;
    bool skip_media = false;  // orc_first_hit only
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override {  // :39-50
        bool any = false;
        HitRecord best{};
        for (const Hittable *obj : objects) {
            if (skip_media && obj->is_medium()) continue;
            HitRecord h;
            if (!obj->hit(r, t_min, t_max, &h)) continue;
            // min_by with a comparator that never returns Equal: keep `best` only if best.t < h.t
            if (!any || !(best.t < h.t)) best = h;
            any = true;
        }
        if (any) *rec = best;
        return any;
    }
    bool bounding_box(Real t0, Real t1, Aabb *out) const override {  // :52-57
        bool any = false;
        for (const Hittable *obj : objects) {
            Aabb b;
            if (!obj->bounding_box(t0, t1, &b)) continue;
            *out = any ? merge_aabbs(*out, b) : b;
            any = true;
        }
        return any;
    }
    Real pdf_value(Vec3 o, Vec3 v) const override {  // :59-69
        if (objects.empty()) return 0.0f;
        Real weight = 1.0f / (Real)objects.size();
        Real sum = 0.0f;
        for (const Hittable *obj : objects) sum = sum + weight * obj->pdf_value(o, v);
        return sum;
    }
    Vec3 random(Vec3 v) const override {  // :71-75
        int num_objects = (int)objects.size() - 1;
        return objects[random_int(0, num_objects)]->random(v);
    }
};

struct Block : Hittable {  // block.rs
    Vec3 box_min, box_max;
    HittableList sides;
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override { return sides.hit(r, t_min, t_max, rec); }
    bool bounding_box(Real, Real, Aabb *out) const override { *out = Aabb{box_min, box_max}; return true; }
};

struct FlipFace : Hittable {  // flip_face.rs
    const Hittable *obj;
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override {
        if (!obj->hit(r, t_min, t_max, rec)) return false;
        rec->front_face = !rec->front_face;
        return true;
    }
    bool bounding_box(Real t0, Real t1, Aabb *out) const override { return obj->bounding_box(t0, t1, out); }
};

/* ------------------------------------------------------------------ transform.rs */
struct Translate : Hittable {  // :14-55
    const Hittable *obj;
    Vec3 offset;
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override {
        Ray translated_ray{r.origin - offset, r.direction, r.time};
        HitRecord h;
        if (!obj->hit(translated_ray, t_min, t_max, &h)) return false;
        *rec = hit_record_new(h.p + offset, h.normal, translated_ray, h.t, h.mtl, h.u, h.v, h.prim_id);
        return true;
    }
    bool bounding_box(Real t0, Real t1, Aabb *out) const override {
        Aabb b;
        if (!obj->bounding_box(t0, t1, &b)) return false;
        *out = Aabb{b.min + offset, b.max + offset};
        return true;
    }
    Real pdf_value(Vec3 o, Vec3 v) const override { return obj->pdf_value(o, v); }
    Vec3 random(Vec3 v) const override { return obj->random(v); }
};

struct RotateY : Hittable {  // :57-153
    const Hittable *obj;
    Real sin_theta, cos_theta;
    Aabb bbox;
    bool init(const Hittable *o, Real angle, const Real *given_sincos = nullptr) {  // :65-108
        obj = o;
        if (given_sincos) { sin_theta = given_sincos[0]; cos_theta = given_sincos[1]; }
        else m_sincos(degrees_to_radians(angle), &sin_theta, &cos_theta);
        Aabb b;
        if (!obj->bounding_box(0.0f, 1.0f, &b)) return false;
        Real mn[3] = {F32_MAX, F32_MAX, F32_MAX}, mx[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    Real x = (Real)i * b.max.x + (1.0f - (Real)i) * b.min.x;
                    Real y = (Real)j * b.max.y + (1.0f - (Real)j) * b.min.y;
                    Real z = (Real)k * b.max.z + (1.0f - (Real)k) * b.min.z;
                    Real tester[3] = {cos_theta * x + sin_theta * z, y, -sin_theta * x + cos_theta * z};
                    for (int c = 0; c < 3; ++c) {
                        if (tester[c] > mx[c]) mx[c] = tester[c];
                        if (tester[c] < mn[c]) mn[c] = tester[c];
                    }
                }
        bbox = Aabb{V(mn[0], mn[1], mn[2]), V(mx[0], mx[1], mx[2])};
        return true;
    }
    bool bounding_box(Real, Real, Aabb *out) const override { *out = bbox; return true; }
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override {  // :116-144
        Vec3 origin = r.origin, direction = r.direction;
        origin.x = cos_theta * r.origin.x - sin_theta * r.origin.z;
        origin.z = sin_theta * r.origin.x + cos_theta * r.origin.z;
        direction.x = cos_theta * r.direction.x - sin_theta * r.direction.z;
        direction.z = sin_theta * r.direction.x + cos_theta * r.direction.z;
        Ray rotated_r{origin, direction, r.time};
        HitRecord h;
        if (!obj->hit(rotated_r, t_min, t_max, &h)) return false;
        Vec3 p = h.p, n = h.normal;
        p.x = cos_theta * h.p.x + sin_theta * h.p.z;
        p.z = -sin_theta * h.p.x + cos_theta * h.p.z;
        n.x = cos_theta * h.normal.x + sin_theta * h.normal.z;
        n.z = -sin_theta * h.normal.x + cos_theta * h.normal.z;
        *rec = h;
        rec->p = p;
        rec->normal = n;
        return true;
    }
    Real pdf_value(Vec3 o, Vec3 v) const override { return obj->pdf_value(o, v); }
    Vec3 random(Vec3 v) const override { return obj->random(v); }
};

/* ------------------------------------------------------------------ math/src/mat4.rs (what Transform uses) */
struct Mat4 {  // row-major a[r*4+c], mat4.rs:20-41
    Real a[16];
    Real at(int r, int c) const { return a[r * 4 + c]; }
    Real &at(int r, int c) { return a[r * 4 + c]; }
};
#define A(r, c) m.at(r, c)
inline Real mat4_det(const Mat4 &m) {  // mat4.rs:536-561, the two index typos in l2 and l4 kept
    Real k1 = A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0), l1 = A(2, 2) * A(3, 3) - A(2, 3) * A(3, 2);
    Real k2 = A(0, 0) * A(1, 2) - A(0, 2) * A(1, 0), l2 = A(2, 1) * A(3, 3) - A(1, 2) * A(3, 1);
    Real k3 = A(0, 0) * A(1, 3) - A(0, 3) * A(1, 0), l3 = A(2, 1) * A(3, 2) - A(2, 2) * A(3, 1);
    Real k4 = A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1), l4 = A(2, 0) * A(3, 3) - A(3, 2) * A(3, 0);
    Real k5 = A(0, 1) * A(1, 3) - A(0, 3) * A(1, 1), l5 = A(2, 0) * A(3, 2) - A(2, 2) * A(3, 0);
    Real k6 = A(0, 2) * A(1, 3) - A(0, 3) * A(1, 2), l6 = A(2, 0) * A(3, 1) - A(2, 1) * A(3, 0);
    return k1 * l1 - k2 * l2 + k3 * l3 + k4 * l4 - k5 * l5 + k6 * l6;
}
inline Mat4 mat4_adjoint(const Mat4 &m) {  // mat4.rs:567-606
    Real m1 = A(2, 2) * A(3, 3) - A(2, 3) * A(3, 2), m2 = A(2, 1) * A(3, 3) - A(2, 3) * A(3, 1);
    Real m3 = A(2, 1) * A(3, 2) - A(2, 2) * A(3, 1), m4 = A(0, 2) * A(1, 3) - A(0, 3) * A(1, 2);
    Real m5 = A(0, 1) * A(1, 3) - A(0, 3) * A(1, 1), m6 = A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1);
    Real m7 = A(2, 0) * A(3, 3) - A(2, 3) * A(3, 0), m8 = A(2, 0) * A(3, 2) - A(2, 2) * A(3, 0);
    Real m9 = A(1, 2) * A(3, 3) - A(1, 3) * A(3, 2), m10 = A(1, 0) * A(3, 3) - A(1, 3) * A(3, 0);
    Real m11 = A(1, 0) * A(3, 2) - A(1, 2) * A(3, 0), m12 = A(0, 0) * A(1, 3) - A(0, 3) * A(1, 0);
    Real m13 = A(0, 0) * A(1, 2) - A(0, 2) * A(1, 0), m14 = A(2, 0) * A(3, 1) - A(2, 1) * A(3, 0);
    Real m15 = A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0), m16 = A(2, 0) * A(3, 1) - A(2, 1) * A(3, 0);
    Mat4 r{};
    r.at(0, 0) = A(1, 1) * m1 - A(1, 2) * m2 + A(1, 3) * m3;   r.at(0, 1) = -A(0, 1) * m1 + A(0, 2) * m2 - A(0, 3) * m3;
    r.at(0, 2) = A(3, 1) * m4 - A(3, 2) * m5 + A(3, 3) * m6;   r.at(0, 3) = -A(2, 1) * m4 + A(2, 2) * m5 - A(2, 3) * m6;
    r.at(1, 0) = -A(1, 0) * m1 + A(1, 2) * m7 - A(1, 3) * m8;  r.at(1, 1) = A(0, 0) * m1 - A(0, 2) * m7 + A(0, 3) * m8;
    r.at(1, 2) = -A(0, 0) * m9 + A(0, 2) * m10 - A(0, 3) * m11; r.at(1, 3) = A(2, 0) * m4 - A(2, 2) * m12 + A(2, 3) * m13;
    r.at(2, 0) = A(1, 0) * m2 - A(1, 1) * m7 + A(1, 3) * m16;  r.at(2, 1) = -A(0, 0) * m2 + A(0, 1) * m7 - A(0, 3) * m16;
    r.at(2, 2) = A(3, 0) * m5 - A(3, 1) * m12 + A(3, 3) * m15; r.at(2, 3) = -A(2, 0) * m5 + A(2, 1) * m12 - A(2, 3) * m15;
    r.at(3, 0) = -A(1, 0) * m3 + A(1, 1) * m8 - A(1, 2) * m14; r.at(3, 1) = A(0, 0) * m3 - A(0, 1) * m8 + A(0, 2) * m14;
    r.at(3, 2) = -A(3, 0) * m6 + A(3, 1) * m13 - A(3, 2) * m15; r.at(3, 3) = A(2, 0) * m6 - A(2, 1) * m13 + A(2, 2) * m15;
    return r;
}
#undef A
inline bool mat4_invert(const Mat4 &m, Mat4 *out) {  // mat4.rs:608-617 (asserts on det == 0)
    Real d = mat4_det(m);
    if (d == 0.0f) return false;
    Real inv = 1.0f / d;
    Mat4 adj = mat4_adjoint(m);
    for (int i = 0; i < 16; ++i) out->a[i] = adj.a[i] * inv;
    return true;
}
inline Mat4 mat4_transpose(const Mat4 &m) {  // mat4.rs:52-74
    Mat4 t{};
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) t.at(r, c) = m.at(c, r);
    return t;
}
inline Vec3 mat4_mul_vec4_xyz(const Mat4 &m, Vec3 v, Real w) {  // mat4.rs:434-441 followed by .xyz()
    return V(m.at(0, 0) * v.x + m.at(0, 1) * v.y + m.at(0, 2) * v.z + m.at(0, 3) * w,
             m.at(1, 0) * v.x + m.at(1, 1) * v.y + m.at(1, 2) * v.z + m.at(1, 3) * w,
             m.at(2, 0) * v.x + m.at(2, 1) * v.y + m.at(2, 2) * v.z + m.at(2, 3) * w);
}
inline Aabb aabb_transform(const Mat4 &mat, const Aabb &b) {  // aabb3.rs:81-97 (vec4 min/max component-wise, xyz kept)
    Aabb r;
    Real lo[3], hi[3];
    for (int row = 0; row < 3; ++row) {
        Real xmin = mat.at(row, 0) * b.min.x, xmax = mat.at(row, 0) * b.max.x;
        Real ymin = mat.at(row, 1) * b.min.y, ymax = mat.at(row, 1) * b.max.y;
        Real zmin = mat.at(row, 2) * b.min.z, zmax = mat.at(row, 2) * b.max.z;
        lo[row] = f32_min(xmin, xmax) + f32_min(ymin, ymax) + f32_min(zmin, zmax) + mat.at(row, 3);
        hi[row] = f32_max(xmin, xmax) + f32_max(ymin, ymax) + f32_max(zmin, zmax) + mat.at(row, 3);
    }
    r.min = V(lo[0], lo[1], lo[2]); r.max = V(hi[0], hi[1], hi[2]);
    return r;
}

struct Transform : Hittable {  // transform.rs:155-213
    Mat4 obj2world, world2object, normal2world;
    const Hittable *obj;
    bool init(const Mat4 &o2w, const Hittable *o) {  // :163-175
        obj = o;
        obj2world = o2w;
        if (!mat4_invert(o2w, &world2object)) return false;
        normal2world = mat4_transpose(mat4_adjoint(world2object));
        return true;
    }
    bool bounding_box(Real t0, Real t1, Aabb *out) const override {
        Aabb b;
        if (!obj->bounding_box(t0, t1, &b)) return false;
        *out = aabb_transform(obj2world, b);
        return true;
    }
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override {  // :184-204 + math/src/ray.rs:42-54
        Ray transformed_ray{mat4_mul_vec4_xyz(world2object, r.origin, 1.0f), mat4_mul_vec4_xyz(world2object, r.direction, 0.0f), r.time};
        HitRecord h;
        if (!obj->hit(transformed_ray, t_min, t_max, &h)) return false;
        *rec = h;
        rec->p = mat4_mul_vec4_xyz(obj2world, h.p, 1.0f);
        rec->normal = normalize(mat4_mul_vec4_xyz(normal2world, h.normal, 0.0f));
        return true;
    }
    Real pdf_value(Vec3 o, Vec3 v) const override { return obj->pdf_value(o, v); }
    Vec3 random(Vec3 v) const override { return obj->random(v); }
};

/* ------------------------------------------------------------------ math/src/polynomial.rs + the quadrics */
inline uint32_t poly_quadratic(Real a, Real b, Real c, Real roots[2]) {  // polynomial.rs:11-39
    if (a == 0.0f) {                     // poly_linear(b, c, roots), :3-9
        if (b == 0.0f) return 0;
        roots[0] = -c / b;
        return 1;
    }
    Real delta = b * b - 4.0f * a * c;
    if (delta == 0.0f) {
        roots[0] = -b / 2.0f * a;        // sic
        roots[1] = -b / 2.0f * a;
        return 2;
    }
    if (!(delta < 0.0f)) {               // is_sign_positive; a NaN delta counts as positive whatever its sign bit:
                                         // which sign an invalid operation gives NaN differs between x86 and the GPU
        Real sgn = b != b ? b : std::copysign(1.0f, b);   // f32::signum
        Real q = -(b + sgn * std::sqrt(delta)) / 2.0f;
        roots[0] = q / a;
        roots[1] = c / q;
        if (roots[0] > roots[1]) std::swap(roots[0], roots[1]);
        return 2;
    }
    return 0;
}
const Real C_TWO_PI = C_PI * 2.0f;       // types.rs:29
inline Real atan2_positive(Real y, Real x, Real add) {  // `phi = y.atan2(x); if phi < 0 { phi += add }`
    Real phi = m_atan2(y, x);
    if (phi < 0.0f) phi = phi + add;
    return phi;
}

struct Cylinder : Hittable {  // cylinder.rs
    Real radius, zmin, zmax, phi_max;
    const Material *mtl;
    uint32_t prim_id;
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override {  // :52-128
        Real a = r.direction.x * r.direction.x + r.direction.y * r.direction.y;
        Real b = 2.0f * (r.direction.x * r.origin.x + r.direction.y * r.origin.y);
        Real c = r.origin.x * r.origin.x + r.origin.y * r.origin.y - radius * radius;
        Real roots[2] = {0.0f, 0.0f};
        if (poly_quadratic(a, b, c, roots) == 0) return false;
        Real t0 = roots[0], t1 = roots[1];
        if (t0 > t_max || t1 < t_min) return false;
        Real thit = t0 < t_min ? t1 : t0;
        if (thit > t_max) return false;
        Vec3 p = r.at(thit);
        Real phi = atan2_positive(p.y, p.x, 2.0f * C_PI);
        if (p.z < zmin || p.z > zmax || phi > phi_max) {
            if (thit == t1) return false;
            thit = t1;
            if (t1 > t_max) return false;
            p = r.at(thit);
            phi = atan2_positive(p.y, p.x, 2.0f * C_PI);
            if (p.z < zmin || p.z > zmax || phi > phi_max) return false;
        }
        Real u = phi / phi_max, v = (p.z - zmin) / (zmax - zmin);
        Vec3 dpdu = V(-phi_max * p.y, phi_max * p.x, 0.0f), dpdv = V(0.0f, 0.0f, zmax - zmin);
        *rec = hit_record_new(p, normalize(cross(dpdv, dpdu)), r, thit, mtl, u, v, prim_id);
        return true;
    }
    bool bounding_box(Real, Real, Aabb *out) const override { *out = Aabb{V(-radius, -radius, zmin), V(radius, radius, zmax)}; return true; }
    Real area() const { return (zmax - zmin) * radius * phi_max; }  // :46-48
    Real pdf_value(Vec3 o, Vec3 v) const override {  // :134-150
        HitRecord h;
        if (!hit(Ray{o, v, 0.0f}, 0.0001f, std::numeric_limits<Real>::infinity(), &h)) return 0.0f;
        Real pdf = (1.0f / area()) / (std::fabs(dot(h.normal, -v)) / length_squared(o - h.p));
        return std::isinf(pdf) ? 0.0f : pdf;
    }
    Vec3 random(Vec3 v) const override {  // :152-161; aabb.center() = (min + max) * 0.5, aabb3.rs:52-54
        Vec3 center = (V(-radius, -radius, zmin) + V(radius, radius, zmax)) * 0.5f;
        Vec3 direction = center - v;
        Real distance_squared = length_squared(direction);
        Onb uvw = onb_from(direction);
        return uvw.local_from_vec(random_to_sphere(radius, distance_squared));
    }
};

struct Cone : Hittable {  // cone.rs
    Real radius, height, phi_max;
    const Material *mtl;
    uint32_t prim_id;
    bool hit(const Ray &r, Real, Real t_max, HitRecord *rec) const override {  // :35-121 (t_min is ignored)
        Real k = radius / height;
        k = k * k;
        Real a = r.direction.x * r.direction.x + r.direction.y * r.direction.y - k * r.direction.z * r.direction.z;
        Real b = 2.0f * (r.direction.x * r.origin.x + r.direction.y * r.origin.y - k * r.direction.z * (r.origin.z - height));
        Real c = r.origin.x * r.origin.x + r.origin.y * r.origin.y - k * (r.origin.z - height) * (r.origin.z - height);
        Real roots[2] = {0.0f, 0.0f};
        if (poly_quadratic(a, b, c, roots) == 0) return false;
        Real t0 = roots[0], t1 = roots[1];
        if (t0 > t_max || t1 <= 0.0f) return false;
        Real thit = t0 <= 0.0f ? t1 : t0;
        if (thit > t_max) return false;
        Vec3 p = r.at(thit);
        Real phi = atan2_positive(p.y, p.x, C_TWO_PI);
        if (p.z < 0.0f || p.z > height || phi > phi_max) {
            if (thit == t1) return false;
            thit = t1;
            if (thit > t_max) return false;
            p = r.at(thit);
            phi = atan2_positive(p.y, p.x, C_TWO_PI);
            if (p.z < 0.0f || p.z > height || phi > phi_max) return false;
        }
        Real u = phi / phi_max, v = p.z / height;
        Vec3 dpdu = V(-phi_max * p.y, phi_max * p.y, 0.0f);  // sic: p.y twice (:109)
        Vec3 dpdv = V(-p.x / (1.0f - v), -p.y / (1.0f - v), height);
        *rec = hit_record_new(p, normalize(cross(dpdu, dpdv)), r, thit, mtl, u, v, prim_id);
        return true;
    }
    bool bounding_box(Real, Real, Aabb *out) const override { *out = Aabb{V(-radius, -radius, 0.0f), V(radius, radius, height)}; return true; }
};

struct Paraboloid : Hittable {  // paraboloid.rs
    Real radius, zmin, zmax, phi_max;
    const Material *mtl;
    uint32_t prim_id;
    bool bounding_box(Real, Real, Aabb *out) const override { *out = Aabb{V(-radius, -radius, zmin), V(radius, radius, zmax)}; return true; }
    bool hit(const Ray &r, Real, Real t_max, HitRecord *rec) const override {  // :48-144 (t_min is ignored)
        Real dx = r.direction.x, dy = r.direction.y, dz = r.direction.z, ox = r.origin.x, oy = r.origin.y, oz = r.origin.z;
        Real k = zmax / (radius * radius);
        Real a = k * (dx * dx + dy * dy);
        Real b = 2.0f * k * (dx * ox + dy * oy) - dz;
        Real c = k * (ox * ox + oy * oy) - oz;
        Real roots[2] = {0.0f, 0.0f};
        if (poly_quadratic(a, b, c, roots) == 0) return false;
        Real t0 = roots[0], t1 = roots[1];
        if (t0 > t_max || t1 <= 0.0f) return false;
        Real thit = t0;
        if (thit <= 0.0f) {
            thit = t1;
            if (thit > t_max) return false;
        }
        Vec3 phit = r.at(thit);
        Real phi = atan2_positive(phit.y, phit.x, C_TWO_PI);
        if (phit.z < zmin || phit.z > zmax || phi > phi_max) {
            if (thit == t1) return false;
            thit = t1;
            if (t1 > t_max) return false;
            phit = r.at(thit);
            phi = atan2_positive(phit.y, phit.x, C_TWO_PI);
            if (phit.z < zmin || phit.z > zmax || phi > phi_max) return false;
        }
        Real u = phi / phi_max, v = (phit.z - zmin) / (zmax - zmin);
        Vec3 dpdu = V(-phi_max * phit.y, phi_max * phit.x, 0.0f);
        Vec3 dpdv = (zmax - zmin) * V(phit.x / (2.0f * phit.z), phit.y / (2.0f * phit.z), 1.0f);
        *rec = hit_record_new(phit, normalize(cross(dpdu, dpdv)), r, thit, mtl, u, v, prim_id);
        return true;
    }
};

struct Hyperboloid : Hittable {  // hyperboloid.rs
    Vec3 p1, p2;
    Real zmin, zmax, phi_max, rmax, ah, ch;
    const Material *mtl;
    uint32_t prim_id;
    bool init(Vec3 a1, Vec3 a2, Real phimax) {  // :26-67
        Real radius1 = std::sqrt(a1.x * a1.x + a1.y * a1.y), radius2 = std::sqrt(a2.x * a2.x + a2.y * a2.y);
        rmax = f32_max(radius1, radius2);
        zmin = f32_min(a1.z, a2.z); zmax = f32_max(a1.z, a2.z);
        p1 = a1; p2 = a2;
        if (a2.z == 0.0f) { p1 = a2; p2 = a1; }
        phi_max = phimax;
        Vec3 pp = p1;
        for (int it = 0; it < 64; ++it) {  // `loop` in the reference; bounded here
            pp = pp + 2.0f * (p2 - p1);
            Real xy1 = pp.x * pp.x + pp.y * pp.y, xy2 = p2.x * p2.x + p2.y * p2.y;
            ah = (1.0f / xy1 - (pp.z * pp.z) / (xy1 * p2.z * p2.z)) / (1.0f - (xy2 * pp.z * pp.z) / (xy1 * p2.z * p2.z));
            ch = (ah * xy2 - 1.0f) / (p2.z * p2.z);
            if (!std::isinf(ah) && !std::isnan(ah)) return true;
        }
        return false;
    }
    bool bounding_box(Real, Real, Aabb *out) const override { *out = Aabb{V(-rmax, -rmax, zmin), V(rmax, rmax, zmax)}; return true; }
    Real phi_of(Vec3 phit, Real *v, Real add) const {  // :112-117, 134-139
        *v = (phit.z - p1.z) / (p2.z - p1.z);
        Vec3 pr = (1.0f - *v) * p1 + *v * p2;
        return atan2_positive(pr.x * phit.y - phit.x * pr.y, phit.x * pr.x + phit.y * pr.y, add);
    }
    bool hit(const Ray &r, Real, Real t_max, HitRecord *rec) const override {  // :78-169 (t_min is ignored)
        Real ox = r.origin.x, oy = r.origin.y, oz = r.origin.z, dx = r.direction.x, dy = r.direction.y, dz = r.direction.z;
        Real a = ah * dx * dx + ah * dy * dy - ch * dz * dz;
        Real b = 2.0f * (ah * dx * ox + ah * dy * oy - ch * dz * oz);
        Real c = ah * ox * ox + ah * oy * oy - ch * oz * oz - 1.0f;
        Real roots[2] = {0.0f, 0.0f};
        if (poly_quadratic(a, b, c, roots) == 0) return false;
        Real t0 = roots[0], t1 = roots[1];
        if (t0 > t_max || t1 <= 0.0f) return false;
        Real thit = t0;
        if (thit <= 0.0f) {
            thit = t1;
            if (thit > t_max) return false;
        }
        Vec3 phit = r.at(thit);
        Real v;
        Real phi = phi_of(phit, &v, 2.0f);  // sic: += C_TWO, not C_TWO_PI (:116)
        if (phit.z < zmin || phit.z > zmax || phi > phi_max) {
            if (thit == t1) return false;
            thit = t1;
            if (t1 > t_max) return false;
            phit = r.at(thit);
            phi = phi_of(phit, &v, C_TWO_PI);
            if (phit.z < zmin || phit.z > zmax || phi > phi_max) return false;
        }
        Real u = phi / phi_max;
        Real sin_phi, cos_phi;
        m_sincos(phi, &sin_phi, &cos_phi);
        Vec3 dpdu = V(-phi_max * phit.y, phi_max * phit.x, 0.0f);
        Vec3 dpdv = V((p2.x - p1.x) * cos_phi - (p2.y - p1.y) * sin_phi, (p2.x - p1.x) * sin_phi + (p2.y - p1.y) * cos_phi, p2.z - p1.z);
        *rec = hit_record_new(phit, normalize(cross(dpdu, dpdv)), r, thit, mtl, u, v, prim_id);
        return true;
    }
};

/* ------------------------------------------------------------------ bvh.rs */
bool g_bvh_exhaustive = false;

struct BvhNode : Hittable {
    const Hittable *left, *right;
    Aabb bbox;
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override {  // :59-84
        if (!g_bvh_exhaustive && !bbox.hit(r, t_min, t_max)) return false;
        HitRecord hl, hr;
        bool has_l = left->hit(r, t_min, t_max, &hl);
        bool has_r = right->hit(r, t_min, has_l ? hl.t : t_max, &hr);
        if (has_l && has_r) { *rec = hl.t < hr.t ? hl : hr; return true; }
        if (has_l) { *rec = hl; return true; }
        if (has_r) { *rec = hr; return true; }
        return false;
    }
    bool bounding_box(Real, Real, Aabb *out) const override { *out = bbox; return true; }
};

/* ------------------------------------------------------------------ constant_medium.rs */
struct ConstantMedium : Hittable {
    const Hittable *boundary;
    const Material *phase_function;
    Real neg_inv_density;
    uint32_t prim_id;
    bool is_medium() const override { return true; }
    bool bounding_box(Real t0, Real t1, Aabb *out) const override { return boundary->bounding_box(t0, t1, out); }
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override {  // :46-94
        HitRecord rec1, rec2;
        if (!boundary->hit(r, -F32_MAX, F32_MAX, &rec1)) return false;
        if (!boundary->hit(r, rec1.t + 0.0001f, F32_MAX, &rec2)) return false;
        if (rec1.t < t_min) rec1.t = t_min;
        if (rec2.t > t_max) rec2.t = t_max;
        if (rec1.t >= rec2.t) return false;
        if (rec1.t < 0.0f) rec1.t = 0.0f;
        Real ray_length = length(r.direction);
        Real distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
        Real hit_distance = neg_inv_density * m_ln(random_real());
        if (hit_distance > distance_inside_boundary) return false;
        Real t = rec1.t + hit_distance / ray_length;
        *rec = HitRecord{r.at(t), V(1.0f, 0.0f, 0.0f), t, phase_function, true, rec1.u, rec1.v, prim_id};
        return true;
    }
};

/* ------------------------------------------------------------------ triangle_mesh.rs */
bool g_mesh_nearest = false;

struct TriangleMesh : Hittable {
    const rtg_mesh *mesh;  // world-space vertices (triangle_mesh.rs:53-72 is done by the host library)
    Aabb aabb;
    std::vector<Aabb> node_aabb;
    const Material *mtl;
    uint32_t first_prim_id;
    std::vector<uint32_t> node_first_tri;
    void init() {  // :51, 64, 79-82
        aabb = aabb_default();
        for (uint32_t i = 0; i < mesh->n_vertices; ++i) aabb.add_point(V(mesh->vertices[i].pos[0], mesh->vertices[i].pos[1], mesh->vertices[i].pos[2]));
        uint32_t tri = 0;
        for (uint32_t n = 0; n < mesh->n_nodes; ++n) {
            Aabb b = aabb_default();
            for (uint32_t k = 0; k < mesh->nodes[n].index_count; ++k) {
                const float *p = mesh->vertices[mesh->indices[mesh->nodes[n].index_begin + k]].pos;
                b.add_point(V(p[0], p[1], p[2]));
            }
            node_aabb.push_back(b);
            node_first_tri.push_back(tri);
            tri += mesh->nodes[n].index_count / 3;
        }
    }
    static Vec3 v3(const float *p) { return V(p[0], p[1], p[2]); }
    bool ray_triangle_intersect_test(const uint32_t *idx, const Ray &r, Real t_min, Real t_max, uint32_t prim_id,
                                     HitRecord *rec) const {  // :185-242
        const rtg_vertex &p1 = mesh->vertices[idx[0]], &p2 = mesh->vertices[idx[1]], &p3 = mesh->vertices[idx[2]];
        Vec3 e1 = v3(p2.pos) - v3(p1.pos), e2 = v3(p3.pos) - v3(p1.pos);
        Vec3 s1 = cross(r.direction, e2);
        Real div = dot(s1, e1);
        if (div == 0.0f) return false;
        Real inv_div = 1.0f / div;
        Vec3 d = r.origin - v3(p1.pos);
        Real b1 = dot(d, s1) * inv_div;
        if (b1 < 0.0f || b1 > 1.0f) return false;
        Vec3 s2 = cross(d, e1);
        Real b2 = dot(r.direction, s2) * inv_div;
        if (b2 < 0.0f || (b1 + b2) > 1.0f) return false;
        Real t = dot(e2, s2) * inv_div;
        if (t < t_min || t > t_max) return false;
        Real b0 = 1.0f - b1 - b2;
        Vec3 n = normalize(b0 * v3(p1.normal) + b1 * v3(p2.normal) + b2 * v3(p3.normal));
        if (dot(r.direction, n) > 0.0f) return false;
        Real u = b0 * p1.uv[0] + b1 * p2.uv[0] + b2 * p3.uv[0];
        Real v = b0 * p1.uv[1] + b1 * p2.uv[1] + b2 * p3.uv[1];
        *rec = hit_record_new(r.at(t), n, r, t, mtl, u, v, prim_id);
        return true;
    }
    bool bounding_box(Real, Real, Aabb *out) const override { *out = aabb; return true; }
    bool hit(const Ray &r, Real t_min, Real t_max, HitRecord *rec) const override {  // :250-267
        if (!g_bvh_exhaustive && !aabb.hit(r, t_min, t_max)) return false;
        bool any = false;
        HitRecord best{};
        for (uint32_t n = 0; n < mesh->n_nodes; ++n) {
            if (!g_bvh_exhaustive && !node_aabb[n].hit(r, t_min, t_max)) continue;
            const uint32_t *ix = mesh->indices + mesh->nodes[n].index_begin;
            uint32_t tris = mesh->nodes[n].index_count / 3;
            bool node_any = false;
            HitRecord node_hit{};
            for (uint32_t k = 0; k < tris; ++k) {
                HitRecord h;
                if (!ray_triangle_intersect_test(ix + 3 * k, r, t_min, t_max, first_prim_id + node_first_tri[n] + k, &h)) continue;
                if (!g_mesh_nearest) { node_hit = h; node_any = true; break; }  // find_map: FIRST in index order
                if (!node_any || !(node_hit.t < h.t)) node_hit = h;            // nearest mode: list semantics
                node_any = true;
            }
            if (!node_any) continue;
            if (!any) best = node_hit;
            else best = best.t < node_hit.t ? best : node_hit;  // :263 reduce
            any = true;
        }
        if (any) *rec = best;
        return any;
    }
};

}  // namespace

/* ------------------------------------------------------------------ scene = object tree rebuilt from the graph */
struct orc_scene {
    std::vector<std::unique_ptr<Texture>> textures;
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Hittable>> objects;
    std::vector<const Hittable *> node_obj;  // graph node -> object (shared nodes are built once)
    HittableList *world = nullptr, *lights = nullptr;
    orc_options opt{};
    uint32_t next_prim = 0;
    rts::Rng bvh_rng;
    const rtg_graph *g = nullptr;
    std::string err;

    template <class T>
    T *own(T *p) { objects.emplace_back(p); return p; }

    // bvh.rs:16-55
    const Hittable *bvh_new(std::vector<const Hittable *> &l, size_t b, size_t e, Real time0, Real time1) {
        int cmp_axis = rts::random_int(bvh_rng, 0, 2);
        auto cmp = [cmp_axis](const Hittable *x, const Hittable *y) {  // :91-100 (Ordering::Less)
            Aabb bx, by;
            x->bounding_box(0.0f, 0.0f, &bx);
            y->bounding_box(0.0f, 0.0f, &by);
            return bx.min[cmp_axis] < by.min[cmp_axis];
        };
        const Hittable *left, *right;
        size_t n = e - b;
        if (n == 1) { left = right = l[b]; }
        else if (n == 2) {
            if (cmp(l[b], l[b + 1])) { left = l[b]; right = l[b + 1]; }
            else { left = l[b + 1]; right = l[b]; }
        } else {
            std::stable_sort(l.begin() + b, l.begin() + e, cmp);  // slice::sort_by is stable
            size_t mid = b + n / 2;
            left = bvh_new(l, b, mid, time0, time1);
            right = bvh_new(l, mid, e, time0, time1);
        }
        Aabb bl, br;
        if (!left->bounding_box(time0, time1, &bl) || !right->bounding_box(time0, time1, &br)) {
            err = "No bounding box in bvh_node constructor";
            return nullptr;
        }
        BvhNode *node = own(new BvhNode());
        node->left = left; node->right = right; node->bbox = merge_aabbs(bl, br);
        return node;
    }

    const Material *mat(int32_t i) {
        if (i < 0 || (size_t)i >= materials.size()) { err = "material index out of range"; return nullptr; }
        return materials[i].get();
    }

    // Primitive ids: depth-first, children in stored order (include/rt_scene_graph.h).  A node
    // reachable twice (the r=70 sphere: world object AND medium boundary) is numbered once, where
    // it is first reached as a world object; a boundary is not numbered.
    const Hittable *build(int32_t idx, bool number, int depth) {
        if (idx < 0 || (uint32_t)idx >= g->n_nodes || depth > 64) { err = "bad node index"; return nullptr; }
        if (node_obj[idx]) return node_obj[idx];
        const rtg_node &n = g->nodes[idx];
        const int32_t *ch = g->children + n.child_begin;
        const Hittable *res = nullptr;
        uint32_t pid = 0xffffffffu;
        auto take_id = [&]() { if (number) pid = next_prim++; };
        switch (n.type) {
        case RTG_SPHERE: {
            take_id();
            Sphere *s = own(new Sphere());
            s->center = V(n.f[0], n.f[1], n.f[2]); s->radius = n.f[3]; s->mtl = mat(n.material); s->prim_id = pid;
            res = s;
            break;
        }
        case RTG_MOVING_SPHERE: {
            take_id();
            MovingSphere *s = own(new MovingSphere());
            s->center0 = V(n.f[0], n.f[1], n.f[2]); s->center1 = V(n.f[3], n.f[4], n.f[5]);
            s->time0 = n.f[6]; s->time1 = n.f[7]; s->radius = n.f[8]; s->mtl = mat(n.material); s->prim_id = pid;
            res = s;
            break;
        }
        case RTG_XY_RECT: case RTG_XZ_RECT: case RTG_YZ_RECT: {
            take_id();
            res = make_rect(n.type, n.f[0], n.f[1], n.f[2], n.f[3], n.f[4], mat(n.material), pid);
            break;
        }
        case RTG_BLOCK: {  // block.rs:22-86
            Block *b = own(new Block());
            const float *p0 = n.f, *p1 = n.f + 3;
            b->box_min = V(p0[0], p0[1], p0[2]); b->box_max = V(p1[0], p1[1], p1[2]);
            const Material *m = mat(n.material);
            auto id = [&]() { return number ? next_prim++ : 0xffffffffu; };
            b->sides.objects.push_back(make_rect(RTG_XY_RECT, p0[0], p1[0], p0[1], p1[1], p1[2], m, id()));
            b->sides.objects.push_back(make_rect(RTG_XY_RECT, p0[0], p1[0], p0[1], p1[1], p0[2], m, id()));
            b->sides.objects.push_back(make_rect(RTG_XZ_RECT, p0[0], p1[0], p0[2], p1[2], p1[1], m, id()));
            b->sides.objects.push_back(make_rect(RTG_XZ_RECT, p0[0], p1[0], p0[2], p1[2], p0[1], m, id()));
            b->sides.objects.push_back(make_rect(RTG_YZ_RECT, p0[1], p1[1], p0[2], p1[2], p1[0], m, id()));
            b->sides.objects.push_back(make_rect(RTG_YZ_RECT, p0[1], p1[1], p0[2], p1[2], p0[0], m, id()));
            res = b;
            break;
        }
        case RTG_FLIP_FACE: {
            const Hittable *c = build(ch[0], number, depth + 1);
            if (!c) return nullptr;
            FlipFace *f = own(new FlipFace());
            f->obj = c;
            res = f;
            break;
        }
        case RTG_TRANSLATE: {
            const Hittable *c = build(ch[0], number, depth + 1);
            if (!c) return nullptr;
            Translate *t = own(new Translate());
            t->obj = c; t->offset = V(n.f[0], n.f[1], n.f[2]);
            res = t;
            break;
        }
        case RTG_ROTATE_Y: {
            const Hittable *c = build(ch[0], number, depth + 1);
            if (!c) return nullptr;
            RotateY *r = own(new RotateY());
            if (!r->init(c, n.f[0], n.i[0] == 1 ? &n.f[1] : nullptr)) { err = "Object does not have a bounding box ..."; return nullptr; }
            res = r;
            break;
        }
        case RTG_TRANSFORM: {
            const Hittable *c = build(ch[0], number, depth + 1);
            if (!c) return nullptr;
            Mat4 o2w{};
            memcpy(o2w.a, n.f, 12 * sizeof(float));
            o2w.at(3, 3) = 1.0f;
            Transform *t = own(new Transform());
            if (!t->init(o2w, c)) { err = "Matrix is not invertible"; return nullptr; }
            res = t;
            break;
        }
        case RTG_CYLINDER: {
            take_id();
            Cylinder *q = own(new Cylinder());
            q->radius = n.f[0]; q->zmin = n.f[1]; q->zmax = n.f[2]; q->phi_max = n.f[3]; q->mtl = mat(n.material); q->prim_id = pid;
            res = q;
            break;
        }
        case RTG_CONE: {
            take_id();
            Cone *q = own(new Cone());
            q->radius = n.f[0]; q->height = n.f[1]; q->phi_max = n.f[2]; q->mtl = mat(n.material); q->prim_id = pid;
            res = q;
            break;
        }
        case RTG_PARABOLOID: {
            take_id();
            Paraboloid *q = own(new Paraboloid());
            q->radius = n.f[0]; q->zmin = n.f[1]; q->zmax = n.f[2]; q->phi_max = n.f[3]; q->mtl = mat(n.material); q->prim_id = pid;
            res = q;
            break;
        }
        case RTG_HYPERBOLOID: {
            take_id();
            Hyperboloid *q = own(new Hyperboloid());
            if (!q->init(V(n.f[0], n.f[1], n.f[2]), V(n.f[3], n.f[4], n.f[5]), n.f[6])) { err = "Hyperboloid without finite coefficients"; return nullptr; }
            q->mtl = mat(n.material); q->prim_id = pid;
            res = q;
            break;
        }
        case RTG_BVH: {
            std::vector<const Hittable *> l;
            for (int i = 0; i < n.child_count; ++i) {
                const Hittable *c = build(ch[i], number, depth + 1);
                if (!c) return nullptr;
                l.push_back(c);
            }
            res = bvh_new(l, 0, l.size(), 0.0f, 1.0f);  // main.rs:715, 807
            break;
        }
        case RTG_LIST: {
            HittableList *l = own(new HittableList());
            for (int i = 0; i < n.child_count; ++i) {
                const Hittable *c = build(ch[i], number, depth + 1);
                if (!c) return nullptr;
                l->objects.push_back(c);
            }
            res = l;
            break;
        }
        case RTG_CONSTANT_MEDIUM: {
            take_id();
            const Hittable *b = build(ch[0], false, depth + 1);
            if (!b) return nullptr;
            ConstantMedium *m = own(new ConstantMedium());
            m->boundary = b; m->phase_function = mat(n.material);
            m->neg_inv_density = -1.0f / n.f[0];  // constant_medium.rs:24
            m->prim_id = pid;
            res = m;
            break;
        }
        case RTG_TRIANGLE_MESH: {
            if (n.i[0] < 0 || (uint32_t)n.i[0] >= g->n_meshes) { err = "mesh index out of range"; return nullptr; }
            TriangleMesh *m = own(new TriangleMesh());
            m->mesh = &g->meshes[n.i[0]];
            m->mtl = mat(n.material);
            m->first_prim_id = next_prim;
            m->init();
            if (number) next_prim += m->mesh->n_indices / 3;
            res = m;
            break;
        }
        default:
            err = "unknown node type";
            return nullptr;
        }
        if (!err.empty()) return nullptr;
        node_obj[idx] = res;
        return res;
    }

    const Hittable *make_rect(int type, Real a0, Real a1, Real b0, Real b1, Real k, const Material *m, uint32_t pid) {
        if (type == RTG_XY_RECT) { XYRect *r = own(new XYRect()); r->a0 = a0; r->a1 = a1; r->b0 = b0; r->b1 = b1; r->k = k; r->mtl = m; r->prim_id = pid; return r; }
        if (type == RTG_XZ_RECT) { XZRect *r = own(new XZRect()); r->a0 = a0; r->a1 = a1; r->b0 = b0; r->b1 = b1; r->k = k; r->mtl = m; r->prim_id = pid; return r; }
        YZRect *r = own(new YZRect()); r->a0 = a0; r->a1 = a1; r->b0 = b0; r->b1 = b1; r->k = k; r->mtl = m; r->prim_id = pid; return r;
    }
};

namespace {

struct GlobalOptions {  // options live in globals read by the object code; set for the duration of a call
    explicit GlobalOptions(const orc_scene *s) {
        g_use_libm = s->opt.use_libm != 0;
        g_mesh_nearest = s->opt.mesh_nearest != 0;
        g_bvh_exhaustive = s->opt.exhaustive != 0;
    }
};

std::atomic<int> g_trace{getenv("ORC_TRACE") != nullptr ? 1 : 0};   // env ORC_TRACE or orc_set_trace()

// camera.rs:253-309
Color ray_color(const Ray &r, Color background, const HittableList &world, const Hittable *lights, int depth) {
    if (depth <= 0) return C(0, 0, 0);
    ++tl_rays;
    HitRecord rec;
    const bool trace = g_trace.load(std::memory_order_relaxed) != 0;   // debugging aid: one line per bounce on stderr
    if (!world.hit(r, 0.001f, INFINITY, &rec)) {
        if (trace) fprintf(stderr, "[orc] depth %d ray %.9g %.9g %.9g | %.9g %.9g %.9g | %.9g miss\n", depth, r.origin.x, r.origin.y, r.origin.z,
                           r.direction.x, r.direction.y, r.direction.z, r.time);
        return background;
    }
    if (trace) fprintf(stderr, "[orc] depth %d ray %.9g %.9g %.9g | %.9g %.9g %.9g | %.9g hit prim %u t %.9g front %d n %.9g %.9g %.9g\n", depth,
                       r.origin.x, r.origin.y, r.origin.z, r.direction.x, r.direction.y, r.direction.z, r.time, rec.prim_id, rec.t,
                       (int)rec.front_face, rec.normal.x, rec.normal.y, rec.normal.z);
    Color emitted = rec.mtl->emitted(r, rec, rec.u, rec.v, rec.p);
    ScatterRecord scatter;
    if (!rec.mtl->scatter(r, rec, &scatter)) return emitted;
    if (scatter.is_specular) return scatter.attenuation * ray_color(scatter.ray, background, world, lights, depth - 1);
    HittablePdf light_pdf(lights, rec.p);
    MixturePdf mixed_pdf(&light_pdf, scatter.pdf.get());
    Ray scattered_ray{rec.p, mixed_pdf.generate(), r.time};
    Real pdf_val = mixed_pdf.value(scattered_ray.direction);
    if (std::fabs(pdf_val) < 1.0E-5f) pdf_val = std::signbit(pdf_val) ? -1.0E-4f : 1.0E-4f;
    Real spdf = rec.mtl->scattering_pdf(r, rec, scattered_ray);
    return emitted + ((scatter.attenuation * spdf) * ray_color(scattered_ray, background, world, lights, depth - 1)) / pdf_val;
}

struct Vec2 { Real x, y; };

// The four get_ray_* of camera.rs:82-184 for a point (u, v) of the view plane and a unit-disk lens sample.
bool project(const orc_render_params &p, Real u, Real v, Vec2 disk, Ray &r) {
    const rt_camera &cam = p.camera;
    Vec3 origin = V(cam.origin[0], cam.origin[1], cam.origin[2]);
    Vec3 llc = V(cam.lower_left_corner[0], cam.lower_left_corner[1], cam.lower_left_corner[2]);
    Vec3 horizontal = V(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    Vec3 vertical = V(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    Vec3 cu = V(cam.u[0], cam.u[1], cam.u[2]), cv = V(cam.v[0], cam.v[1], cam.v[2]), cw = V(cam.w[0], cam.w[1], cam.w[2]);
    if (p.projection == RT_PROJ_PERSPECTIVE || p.projection == RT_PROJ_ORTHOGRAPHIC) {
        Vec2 rd{cam.lens_radius * disk.x, cam.lens_radius * disk.y};
        Vec3 offset = cu * rd.x + cv * rd.y;
        if (p.projection == RT_PROJ_PERSPECTIVE) {  // camera.rs:82-96
            r.origin = origin + offset;
            r.direction = llc + u * horizontal + v * vertical - origin - offset;
        } else {                                    // camera.rs:98-112
            r.origin = llc + u * horizontal + v * vertical + offset;
            r.direction = -cw;
        }
        return true;
    }
    Vec2 pn{2.0f * u - 1.0f, 2.0f * v - 1.0f};
    r.origin = origin;
    if (p.projection == RT_PROJ_FISHEYE) {          // camera.rs:114-148
        Real rsqr = pn.x * pn.x + pn.y * pn.y;
        if (!(rsqr <= 1.0f)) return false;
        Real rr = std::sqrt(rsqr);
        Real psi = rr * p.psi_max;
        Real sin_psi, cos_psi;
        m_sincos(psi, &sin_psi, &cos_psi);
        Real cos_alpha = pn.x / rr, sin_alpha = pn.y / rr;
        r.direction = (sin_psi * cos_alpha) * cu + (sin_psi * sin_alpha) * cv - cos_psi * cw;
    } else {                                        // camera.rs:150-184
        Real lambda = pn.x * p.lambda_max, psi = pn.y * p.psi_max;
        Real phi = 3.14159265358979323846f - lambda;
        Real theta = 0.5f * 3.14159265358979323846f - psi;
        Real sin_theta, cos_theta, sin_phi, cos_phi;
        m_sincos(theta, &sin_theta, &cos_theta);
        m_sincos(phi, &sin_phi, &cos_phi);
        r.direction = (sin_theta * sin_phi) * cu + cos_theta * cv + (sin_theta * cos_phi) * cw;
    }
    return true;
}

// camera.rs:186-251 for one pixel, samples [spp_begin, spp_begin + spp_count)
Color raytrace_pixel(int x, int y, const orc_render_params &p, const rt_sampler_desc &smp, const HittableList &world,
                     const Hittable *lights) {
    Vec3 origin = V(p.camera.origin[0], p.camera.origin[1], p.camera.origin[2]);
    Color background = C(p.background[0], p.background[1], p.background[2]);
    uint32_t pixel = (uint32_t)y * (uint32_t)p.image_width + (uint32_t)x;
    // Perspective / Orthographic draw a lens sample too: the square and disk look-ups share the sampler's counter
    // (sampling.rs:137-163), so a sample takes slots 2s and 2s+1; the other projections take slot s.
    const bool lens = p.projection == RT_PROJ_PERSPECTIVE || p.projection == RT_PROJ_ORTHOGRAPHIC;
    Color color = C(0, 0, 0);
    for (int s = p.spp_begin; s < p.spp_begin + p.spp_count; ++s) {
        rts::Rng rng;
        rts::path_rng(rng, p.seed, pixel, (uint32_t)s);
        tl_rng = &rng;
        uint32_t i0 = rts::sampler_index(p.seed, pixel, lens ? 2u * (uint32_t)s : (uint32_t)s, smp.sets, smp.samples_in_set, smp.shuffled_indices);
        Vec2 off{smp.samples[2 * i0], smp.samples[2 * i0 + 1]};  // camera.rs:196
        Real u = ((Real)x + off.x) / (Real)(p.image_width - 1);
        Real v = 1.0f - ((Real)y + off.y) / (Real)(p.image_height - 1);
        Vec2 disk{0.0f, 0.0f};
        if (lens) {
            uint32_t i1 = rts::sampler_index(p.seed, pixel, 2u * (uint32_t)s + 1u, smp.sets, smp.samples_in_set, smp.shuffled_indices);
            disk = Vec2{smp.disk_samples[2 * i1], smp.disk_samples[2 * i1 + 1]};
        }
        Ray r{origin, origin, 0.0f};
        if (!project(p, u, v, disk, r)) continue;   // FishEye None: the colour is unchanged and no time is drawn
        r.time = random_real();
        color = color + ray_color(r, background, world, lights, p.max_ray_depth);
    }
    tl_rng = nullptr;
    return color;
}

}  // namespace

extern "C" {

const char *orc_last_error(void) { return g_error.c_str(); }

// KAT hook: the ray of get_ray_<projection>(u, v) with lens sample (dx, dy); out6 = origin, direction; 0 = None
void orc_set_trace(int on) { g_trace.store(on ? 1 : 0); }

int orc_project(const orc_render_params *p, float u, float v, float dx, float dy, float *out6) {
    Ray r{V(0, 0, 0), V(0, 0, 0), 0.0f};
    if (!project(*p, u, v, Vec2{dx, dy}, r)) return 0;
    out6[0] = r.origin.x; out6[1] = r.origin.y; out6[2] = r.origin.z;
    out6[3] = r.direction.x; out6[4] = r.direction.y; out6[5] = r.direction.z;
    return 1;
}

int orc_scene_create(const rtg_graph *g, const orc_options *opt, orc_scene **out) {
    if (!g || !out) { set_error("null argument"); return RT_E_INVALID; }
    auto s = std::make_unique<orc_scene>();
    if (opt) s->opt = *opt;
    s->g = g;
    rts::host_rng(s->bvh_rng, s->opt.bvh_seed, 0xb7e15163u);
    // textures (children may come later in the table: two passes)
    s->textures.resize(g->n_textures);
    for (uint32_t i = 0; i < g->n_textures; ++i) {
        const rtg_texture &t = g->textures[i];
        switch (t.type) {
        case RTG_TEX_SOLID: s->textures[i].reset(new SolidColorTexture(C(t.color[0], t.color[1], t.color[2]))); break;
        case RTG_TEX_CHECKER: s->textures[i].reset(new CheckerTexture()); break;
        case RTG_TEX_NOISE: {
            if (t.perlin < 0 || (uint32_t)t.perlin >= g->n_perlins) { set_error("perlin index out of range"); return RT_E_INVALID; }
            NoiseTexture *n = new NoiseTexture();
            n->perlin.tab = &g->perlins[t.perlin];
            n->scale = t.scale;
            s->textures[i].reset(n);
            break;
        }
        case RTG_TEX_IMAGE: {
            ImageTexture *im = new ImageTexture();
            if (t.image >= 0) {
                if ((uint32_t)t.image >= g->n_images) { delete im; set_error("image index out of range"); return RT_E_INVALID; }
                im->width = (uint32_t)g->images[t.image].width;
                im->height = (uint32_t)g->images[t.image].height;
                im->bytes_per_scanline = im->width * 4;
                im->pixels = g->images[t.image].rgba;
            }
            s->textures[i].reset(im);
            break;
        }
        default: set_error("unknown texture type"); return RT_E_INVALID;
        }
    }
    for (uint32_t i = 0; i < g->n_textures; ++i)
        if (g->textures[i].type == RTG_TEX_CHECKER) {
            const rtg_texture &t = g->textures[i];
            if (t.odd < 0 || t.even < 0 || (uint32_t)t.odd >= g->n_textures || (uint32_t)t.even >= g->n_textures) { set_error("checker child out of range"); return RT_E_INVALID; }
            CheckerTexture *c = static_cast<CheckerTexture *>(s->textures[i].get());
            c->odd = s->textures[t.odd].get(); c->even = s->textures[t.even].get(); c->repeat_factor = t.scale;
        }
    auto tex = [&](int32_t i) -> const Texture * { return (i >= 0 && (uint32_t)i < g->n_textures) ? s->textures[i].get() : nullptr; };
    for (uint32_t i = 0; i < g->n_materials; ++i) {
        const rtg_material &m = g->materials[i];
        Material *mm = nullptr;
        switch (m.type) {
        case RTG_MAT_LAMBERTIAN: { auto *x = new Lambertian(); x->albedo = tex(m.tex); mm = x; if (!x->albedo) { delete x; mm = nullptr; } break; }
        case RTG_MAT_METAL: { auto *x = new Metal(); x->albedo = C(m.color[0], m.color[1], m.color[2]); x->fuzziness = m.fuzz; mm = x; break; }
        case RTG_MAT_DIELECTRIC: { auto *x = new Dielectric(); x->refraction_index = m.ior; mm = x; break; }
        case RTG_MAT_DIFFUSE_LIGHT: { auto *x = new DiffuseLight(); x->emit = tex(m.tex); mm = x; if (!x->emit) { delete x; mm = nullptr; } break; }
        case RTG_MAT_ISOTROPIC: { auto *x = new Isotropic(); x->albedo = tex(m.tex); mm = x; if (!x->albedo) { delete x; mm = nullptr; } break; }
        default: break;
        }
        if (!mm) { set_error("bad material %u", i); return RT_E_INVALID; }
        s->materials.emplace_back(mm);
    }
    s->node_obj.assign(g->n_nodes, nullptr);
    const Hittable *w = s->build(g->world_root, true, 0);
    if (!w) { set_error("world: %s", s->err.c_str()); return RT_E_INVALID; }
    const Hittable *l = s->build(g->lights_root, false, 0);
    if (!l) { set_error("lights: %s", s->err.c_str()); return RT_E_INVALID; }
    s->world = const_cast<HittableList *>(static_cast<const HittableList *>(w));
    s->lights = const_cast<HittableList *>(static_cast<const HittableList *>(l));
    *out = s.release();
    return RT_OK;
}

void orc_scene_destroy(orc_scene *s) { delete s; }

int orc_render(const orc_scene *s, const orc_render_params *p, const rt_sampler_desc *smp, float *out, uint64_t *rays_out,
               uint32_t *pixel_rays) {
    if (!s || !p || !smp || !out) { set_error("null argument"); return RT_E_INVALID; }
    if (p->image_width < 2 || p->image_height < 2 || p->spp_count < 0 || p->spp_begin < 0 ||
        p->spp_begin + p->spp_count > p->samples_per_pixel || (int32_t)smp->samples_in_set != p->samples_per_pixel) {
        set_error("bad render params");
        return RT_E_INVALID;
    }
    GlobalOptions go(s);
    int x0 = std::max(0, p->tile_x0), y0 = std::max(0, p->tile_y0);
    int x1 = std::min(p->image_width, p->tile_x1), y1 = std::min(p->image_height, p->tile_y1);
    // main.rs:1265-1282 work blocks over the tile; LIFO pop under a mutex (main.rs:1344-1348)
    struct Blk { int xa, xb, ya, yb; };
    std::vector<Blk> blocks;
    int B = p->block_pixels > 0 ? p->block_pixels : 8;
    for (int by = y0; by < y1; by += B)
        for (int bx = x0; bx < x1; bx += B) blocks.push_back(Blk{bx, std::min(bx + B, x1), by, std::min(by + B, y1)});
    std::mutex mu;
    std::atomic<uint64_t> total_rays{0};
    auto worker = [&]() {
        for (;;) {
            Blk b;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (blocks.empty()) break;
                b = blocks.back();
                blocks.pop_back();
            }
            for (int y = b.yb - 1; y >= b.ya; --y)  // main.rs:1353-1356: rows reversed
                for (int x = b.xa; x < b.xb; ++x) {
                    tl_rays = 0;
                    Color c = raytrace_pixel(x, y, *p, *smp, *s->world, s->lights);
                    float *px = out + ((size_t)y * p->image_width + x) * 4;
                    px[0] += c.r; px[1] += c.g; px[2] += c.b; px[3] += (float)p->spp_count;
                    if (pixel_rays) pixel_rays[(size_t)y * p->image_width + x] += (uint32_t)tl_rays;
                    total_rays += tl_rays;
                }
        }
    };
    int nt = std::max(1, p->threads);
    std::vector<std::thread> th;
    for (int i = 1; i < nt; ++i) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
    if (rays_out) *rays_out = total_rays.load();
    return RT_OK;
}

int orc_first_hit(const orc_scene *s, const float *rays, int32_t n, float t_min, float t_max, uint32_t *prim_id, float *t,
                  float *normal3, uint8_t *front_face, float *uv2) {
    if (!s || !rays || n < 0) { set_error("null argument"); return RT_E_INVALID; }
    GlobalOptions go(s);
    s->world->skip_media = true;
    for (int32_t i = 0; i < n; ++i) {
        Ray r{V(rays[i], rays[(size_t)n + i], rays[2 * (size_t)n + i]),
              V(rays[3 * (size_t)n + i], rays[4 * (size_t)n + i], rays[5 * (size_t)n + i]), rays[6 * (size_t)n + i]};
        HitRecord rec;
        bool h = s->world->hit(r, t_min, t_max, &rec);
        if (prim_id) prim_id[i] = h ? rec.prim_id : RT_MISS;
        if (t) t[i] = h ? rec.t : 0.0f;
        if (normal3) { normal3[3 * i] = h ? rec.normal.x : 0.0f; normal3[3 * i + 1] = h ? rec.normal.y : 0.0f; normal3[3 * i + 2] = h ? rec.normal.z : 0.0f; }
        if (front_face) front_face[i] = h ? (rec.front_face ? 1 : 0) : 0;
        if (uv2) { uv2[2 * i] = h ? rec.u : 0.0f; uv2[2 * i + 1] = h ? rec.v : 0.0f; }
    }
    s->world->skip_media = false;
    return RT_OK;
}

void orc_resolve(const float *sum, int32_t n_pixels, int32_t spp, float *out) {  // main.rs:1363-1388
    Real gamma_correct = 1.0f / (Real)spp;
    for (int32_t i = 0; i < n_pixels; ++i) {
        for (int c = 0; c < 3; ++c) {
            Real x = sum[4 * i + c];
            out[4 * i + c] = !pm::is_normal_(x) ? 0.0f : f32_clamp(std::sqrt(x * gamma_correct), 0.0f, 1.0f);
        }
        out[4 * i + 3] = sum[4 * i + 3];
    }
}

// main.rs:1543-1551: ImageRgb32F -> to_rgb8.  image 0.24.6 (Cargo.lock; not under /root/reference) converts a f32 sample with
// `(float.clamp(0.0, 1.0) * u8::MAX as f32).round()` (traits.rs, FromPrimitive<f32> for u8); f32::round is half away from zero.
void orc_resolve_rgb8(const float *sum, int32_t n_pixels, int32_t spp, uint8_t *out) {
    Real gamma_correct = 1.0f / (Real)spp;
    for (int32_t i = 0; i < n_pixels; ++i)
        for (int c = 0; c < 3; ++c) {
            Real x = sum[4 * i + c];
            Real v = !pm::is_normal_(x) ? 0.0f : f32_clamp(std::sqrt(x * gamma_correct), 0.0f, 1.0f);
            out[3 * i + c] = (uint8_t)std::round(f32_clamp(v, 0.0f, 1.0f) * 255.0f);
        }
}

int orc_texture_value(const orc_scene *s, int32_t tex, float u, float v, const float p[3], float out_rgb[3]) {
    if (!s || tex < 0 || (size_t)tex >= s->textures.size()) return RT_E_INVALID;
    GlobalOptions go(s);
    Color c = s->textures[tex]->value(u, v, V(p[0], p[1], p[2]));
    out_rgb[0] = c.r; out_rgb[1] = c.g; out_rgb[2] = c.b;
    return RT_OK;
}

int orc_aabb_hit(const float mn[3], const float mx[3], const float ray7[7], float t_min, float t_max) {
    Aabb b{V(mn[0], mn[1], mn[2]), V(mx[0], mx[1], mx[2])};
    Ray r{V(ray7[0], ray7[1], ray7[2]), V(ray7[3], ray7[4], ray7[5]), ray7[6]};
    return b.hit(r, t_min, t_max) ? 1 : 0;
}

void orc_onb(const float n[3], float out[9]) {
    Onb o = onb_from(V(n[0], n[1], n[2]));
    for (int i = 0; i < 3; ++i) { out[3 * i] = o.axis[i].x; out[3 * i + 1] = o.axis[i].y; out[3 * i + 2] = o.axis[i].z; }
}

float orc_schlick(float cosine, float ior) { return schlick(cosine, ior); }

void orc_refract(const float uv[3], const float n[3], float eta, float out[3]) {
    Vec3 r = refract(V(uv[0], uv[1], uv[2]), V(n[0], n[1], n[2]), eta);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

float orc_math(int32_t fn, float a, float b, int32_t use_libm) {
    bool saved = g_use_libm;
    g_use_libm = use_libm != 0;
    float r = fn == 0 ? m_sin(a) : fn == 1 ? m_cos(a) : fn == 2 ? m_acos(a) : fn == 3 ? m_atan2(a, b) : m_ln(a);
    g_use_libm = saved;
    return r;
}

float orc_perlin(const rtg_perlin *table, const float p[3], int32_t depth) {
    PerlinNoise n{table};
    return depth > 0 ? n.turbulence(V(p[0], p[1], p[2]), depth) : n.noise(V(p[0], p[1], p[2]));
}

// Known-answer hook: ONE bounce of ray_color (camera.rs:264-308) for a given ray, with the random numbers of that bounce taken
// from xi[0..n_xi) in the order the reference draws them (media in world order, then the material, then the mixture pdf).
// out[0] kind: 0 miss, 1 no scatter (emitted only), 2 specular, 3 pdf;  out[1] prim id  out[2] t  out[3..5] p  out[6..8] normal
// out[9] front_face  out[10..12] scattered direction  out[13..15] attenuation  out[16] mixture pdf value after the clamp of
// camera.rs:279-288  out[17] scattering_pdf  out[18..20] emitted  out[21] numbers drawn.  Returns 0, or RT_E_INVALID.
int orc_bounce(const orc_scene *s, const float ray7[7], const float *xi, int32_t n_xi, float out[22]) {
    if (!s || !ray7 || !out || n_xi < 0) return RT_E_INVALID;
    GlobalOptions go(s);
    for (int i = 0; i < 22; ++i) out[i] = 0.0f;
    static const float none = 0.0f;
    tl_forced = xi ? xi : &none; tl_forced_left = xi ? n_xi : 0; tl_forced_used = 0;
    Ray r{V(ray7[0], ray7[1], ray7[2]), V(ray7[3], ray7[4], ray7[5]), ray7[6]};
    HitRecord rec;
    auto done = [&]() { out[21] = (float)tl_forced_used; tl_forced = nullptr; tl_forced_left = 0; return RT_OK; };
    if (!s->world->hit(r, 0.001f, INFINITY, &rec)) return done();
    out[1] = (float)rec.prim_id; out[2] = rec.t;
    out[3] = rec.p.x; out[4] = rec.p.y; out[5] = rec.p.z;
    out[6] = rec.normal.x; out[7] = rec.normal.y; out[8] = rec.normal.z;
    out[9] = rec.front_face ? 1.0f : 0.0f;
    Color emitted = rec.mtl->emitted(r, rec, rec.u, rec.v, rec.p);
    out[18] = emitted.r; out[19] = emitted.g; out[20] = emitted.b;
    ScatterRecord scatter;
    if (!rec.mtl->scatter(r, rec, &scatter)) { out[0] = 1.0f; return done(); }
    out[13] = scatter.attenuation.r; out[14] = scatter.attenuation.g; out[15] = scatter.attenuation.b;
    if (scatter.is_specular) {
        out[0] = 2.0f;
        out[10] = scatter.ray.direction.x; out[11] = scatter.ray.direction.y; out[12] = scatter.ray.direction.z;
        return done();
    }
    out[0] = 3.0f;
    HittablePdf light_pdf(s->lights, rec.p);
    MixturePdf mixed_pdf(&light_pdf, scatter.pdf.get());
    Ray scattered_ray{rec.p, mixed_pdf.generate(), r.time};
    Real pdf_val = mixed_pdf.value(scattered_ray.direction);
    if (std::fabs(pdf_val) < 1.0E-5f) pdf_val = std::signbit(pdf_val) ? -1.0E-4f : 1.0E-4f;
    out[10] = scattered_ray.direction.x; out[11] = scattered_ray.direction.y; out[12] = scattered_ray.direction.z;
    out[16] = pdf_val;
    out[17] = rec.mtl->scattering_pdf(r, rec, scattered_ray);
    return done();
}

// Known-answer hook for the oracle's own matrix code (math/src/mat4.rs): op 0 transpose (:52-74), 1 invert (:608-617; returns
// RT_E_INVALID for det == 0), 2 adjoint (:567-606).  Row-major 4x4.
int orc_mat4(int32_t op, const float in[16], float out[16]) {
    Mat4 m{}, r{};
    for (int i = 0; i < 16; ++i) m.a[i] = in[i];
    if (op == 0) r = mat4_transpose(m);
    else if (op == 1) { if (!mat4_invert(m, &r)) return RT_E_INVALID; }
    else if (op == 2) r = mat4_adjoint(m);
    else return RT_E_INVALID;
    for (int i = 0; i < 16; ++i) out[i] = r.a[i];
    return RT_OK;
}

float orc_lights_pdf_value(const orc_scene *s, const float o[3], const float v[3]) {
    GlobalOptions go(s);
    return s->lights->pdf_value(V(o[0], o[1], o[2]), V(v[0], v[1], v[2]));
}

}  // extern "C"
