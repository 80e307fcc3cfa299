// rt_host_params.cpp — config (RON), RaytracerParams, camera block, sampler table, work blocks.
//
// Restates raytracer/src/main.rs:1135-1216 + :1241-1282 (config structs, load_config, scene
// lookup, image_height, work blocks), camera.rs:39-80 (Camera::new) and sampling.rs:30-135,
// 233-253 (SamplerBase::new over the multi-jittered strategy).
#include <algorithm>
#include <fstream>
#include <map>
#include <sstream>

#include "rt_host_internal.h"
#include <cctype>
#include <string>

using namespace rth;

/* ------------------------------------------------------------------ RON subset parser */
// Grammar needed by data/config/raytracer.config.ron: structs `(name: value, ...)`, tuples
// `(v, v, ...)`, lists `[v, ...]`, `Some(v)`, `None`, identifiers, numbers, booleans,
// trailing commas, `//` and `/* */` comments.

namespace {

struct Val {
    enum Kind { Num, Ident, Tuple, Struct, List, Some } kind = Ident;
    double num = 0;
    std::string ident;
    std::vector<Val> items;                         // Tuple / List / Some(1 item)
    std::vector<std::pair<std::string, Val>> fields;  // Struct
};

struct Parser {
    const char *p, *end;
    std::string err;

    void skip() {
        for (;;) {
            while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
            if (p + 1 < end && p[0] == '/' && p[1] == '/') {
                while (p < end && *p != '\n') ++p;
            } else if (p + 1 < end && p[0] == '/' && p[1] == '*') {
                p += 2;
                while (p + 1 < end && !(p[0] == '*' && p[1] == '/')) ++p;
                p = (p + 1 < end) ? p + 2 : end;
            } else {
                return;
            }
        }
    }
    bool fail(const std::string &m) {
        if (err.empty()) err = m;
        return false;
    }
    static bool ident_start(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
    static bool ident_char(char c) { return ident_start(c) || (c >= '0' && c <= '9'); }

    bool parse_ident(std::string *out) {
        skip();
        if (p >= end || !ident_start(*p)) return false;
        const char *b = p;
        while (p < end && ident_char(*p)) ++p;
        out->assign(b, p);
        return true;
    }

    // After '(' : struct if the first token is `ident :`.
    bool parse_paren(Val *v) {
        skip();
        const char *save = p;
        std::string id;
        bool is_struct = false;
        if (parse_ident(&id)) {
            skip();
            if (p < end && *p == ':') is_struct = true;
        }
        p = save;
        if (is_struct) {
            v->kind = Val::Struct;
            for (;;) {
                skip();
                if (p < end && *p == ')') { ++p; return true; }
                std::string name;
                if (!parse_ident(&name)) return fail("expected field name");
                skip();
                if (p >= end || *p != ':') return fail("expected ':' after field " + name);
                ++p;
                Val fv;
                if (!parse_value(&fv)) return false;
                v->fields.emplace_back(name, std::move(fv));
                skip();
                if (p < end && *p == ',') ++p;
                else if (p < end && *p == ')') { ++p; return true; }
                else return fail("expected ',' or ')' in struct");
            }
        }
        v->kind = Val::Tuple;
        for (;;) {
            skip();
            if (p < end && *p == ')') { ++p; return true; }
            Val iv;
            if (!parse_value(&iv)) return false;
            v->items.push_back(std::move(iv));
            skip();
            if (p < end && *p == ',') ++p;
            else if (p < end && *p == ')') { ++p; return true; }
            else return fail("expected ',' or ')' in tuple");
        }
    }

    bool parse_value(Val *v) {
        skip();
        if (p >= end) return fail("unexpected end of input");
        char c = *p;
        if (c == '(') { ++p; return parse_paren(v); }
        if (c == '[') {
            ++p;
            v->kind = Val::List;
            for (;;) {
                skip();
                if (p < end && *p == ']') { ++p; return true; }
                Val iv;
                if (!parse_value(&iv)) return false;
                v->items.push_back(std::move(iv));
                skip();
                if (p < end && *p == ',') ++p;
                else if (p < end && *p == ']') { ++p; return true; }
                else return fail("expected ',' or ']' in list");
            }
        }
        if (c == '-' || c == '+' || c == '.' || (c >= '0' && c <= '9')) {
            char *e = nullptr;
            v->kind = Val::Num;
            v->num = strtod(p, &e);
            if (e == p) return fail("bad number");
            p = e;
            return true;
        }
        std::string id;
        if (!parse_ident(&id)) return fail(std::string("unexpected character '") + c + "'");
        skip();
        if (p < end && *p == '(') {  // Some(x) or a named struct `Name(...)`
            ++p;
            Val inner;
            if (!parse_paren(&inner)) return false;
            if (id == "Some") {
                v->kind = Val::Some;
                if (inner.kind == Val::Tuple && inner.items.size() == 1) v->items.push_back(std::move(inner.items[0]));
                else v->items.push_back(std::move(inner));
            } else {
                *v = std::move(inner);
            }
            return true;
        }
        v->kind = Val::Ident;
        v->ident = id;
        return true;
    }
};

const Val *field(const Val &s, const char *name) {
    for (auto &f : s.fields)
        if (f.first == name) return &f.second;
    return nullptr;
}

bool get_num(const Val &s, const char *name, double *out, std::string *err) {
    const Val *v = field(s, name);
    if (!v || v->kind != Val::Num) { *err = std::string("missing or non-numeric field ") + name; return false; }
    *out = v->num;
    return true;
}

bool get_vec3(const Val &s, const char *name, float out[3], std::string *err) {
    const Val *v = field(s, name);
    if (!v || v->kind != Val::Tuple || v->items.size() != 3) { *err = std::string("field ") + name + " is not a 3-tuple"; return false; }
    for (int i = 0; i < 3; ++i) {
        if (v->items[i].kind != Val::Num) { *err = std::string("field ") + name + " holds a non-number"; return false; }
        out[i] = (float)v->items[i].num;
    }
    return true;
}

bool to_user_config(const Val &s, rth_user_config *c, std::string *err) {
    if (s.kind != Val::Struct) { *err = "params is not a struct"; return false; }
    double d;
#define NUM_I(f) if (!get_num(s, #f, &d, err)) return false; c->f = (int32_t)d;
#define NUM_F(f) if (!get_num(s, #f, &d, err)) return false; c->f = (float)d;
    NUM_I(workers) NUM_I(worker_block_pixels) NUM_F(aspect_ratio) NUM_I(image_width) NUM_I(samples_per_pixel)
    NUM_I(max_ray_depth) NUM_F(vertical_fov) NUM_F(aperture) NUM_F(focus_dist) NUM_F(psi_max) NUM_F(lambda_max)
#undef NUM_I
#undef NUM_F
    if (!get_vec3(s, "look_from", c->look_from, err) || !get_vec3(s, "look_at", c->look_at, err) ||
        !get_vec3(s, "world_up", c->world_up, err) || !get_vec3(s, "background", c->background, err))
        return false;
    const Val *sh = field(s, "shuffle_workblocks");
    if (!sh || sh->kind != Val::Ident || (sh->ident != "true" && sh->ident != "false")) { *err = "shuffle_workblocks is not a bool"; return false; }
    c->shuffle_workblocks = sh->ident == "true";
    const Val *pr = field(s, "projection");
    if (!pr || pr->kind != Val::Ident) { *err = "projection missing"; return false; }
    static const char *names[4] = {"Perspective", "Orthographic", "FishEye", "SphericalPanoramic"};  // camera.rs:17-22
    c->projection = -1;
    for (int i = 0; i < 4; ++i)
        if (pr->ident == names[i]) c->projection = i;
    if (c->projection < 0) { *err = "unknown projection " + pr->ident; return false; }
    return true;
}

}  // namespace

struct rth_config {
    std::string active_scene;
    rth_user_config defaults{};
    std::vector<std::pair<std::string, std::pair<bool, rth_user_config>>> scenes;  // (name, (has_params, params))
};

extern "C" {

int rth_config_parse(const char *text, rth_config **out) {
    if (!text || !out) { set_error("null argument"); return RT_E_INVALID; }
    Parser ps{text, text + strlen(text), {}};
    Val root;
    if (!ps.parse_value(&root)) { set_error("config: %s", ps.err.c_str()); return RT_E_INVALID; }
    ps.skip();
    if (ps.p != ps.end) { set_error("config: trailing characters"); return RT_E_INVALID; }
    if (root.kind != Val::Struct) { set_error("config: top level is not a struct"); return RT_E_INVALID; }
    auto cfg = std::make_unique<rth_config>();
    std::string err;
    const Val *as = field(root, "active_scene");
    if (!as || as->kind != Val::Ident) { set_error("config: active_scene missing"); return RT_E_INVALID; }
    cfg->active_scene = as->ident;
    const Val *dp = field(root, "default_params");
    if (!dp || !to_user_config(*dp, &cfg->defaults, &err)) { set_error("config: default_params: %s", err.c_str()); return RT_E_INVALID; }
    const Val *ds = field(root, "defined_scenes");
    if (!ds || ds->kind != Val::List) { set_error("config: defined_scenes missing"); return RT_E_INVALID; }
    for (const Val &e : ds->items) {
        if (e.kind != Val::Tuple || e.items.size() != 2 || e.items[0].kind != Val::Ident) {
            set_error("config: defined_scenes entry is not (Scene, Option<params>)");
            return RT_E_INVALID;
        }
        std::pair<bool, rth_user_config> pr{false, {}};
        const Val &opt = e.items[1];
        if (opt.kind == Val::Some && opt.items.size() == 1) {
            pr.first = true;
            if (!to_user_config(opt.items[0], &pr.second, &err)) {
                set_error("config: scene %s: %s", e.items[0].ident.c_str(), err.c_str());
                return RT_E_INVALID;
            }
        } else if (!(opt.kind == Val::Ident && opt.ident == "None")) {
            set_error("config: scene %s: expected Some(..) or None", e.items[0].ident.c_str());
            return RT_E_INVALID;
        }
        cfg->scenes.emplace_back(e.items[0].ident, pr);
    }
    *out = cfg.release();
    return RT_OK;
}

int rth_config_load(const char *path, rth_config **out) {
    if (!path) { set_error("null path"); return RT_E_INVALID; }
    std::ifstream f(path);
    if (!f) { set_error("Failed to open config file %s", path); return RT_E_INVALID; }
    std::stringstream ss;
    ss << f.rdbuf();
    return rth_config_parse(ss.str().c_str(), out);
}

void rth_config_free(rth_config *c) { delete c; }

const char *rth_config_active_scene(const rth_config *c) { return c ? c->active_scene.c_str() : ""; }

int rth_config_scene_params(const rth_config *c, const char *scene, rth_user_config *out) {
    if (!c || !out) { set_error("null argument"); return RT_E_INVALID; }
    std::string name = scene ? scene : c->active_scene;
    for (auto &s : c->scenes)  // main.rs:1251-1261: first match wins
        if (s.first == name) {
            *out = s.second.first ? s.second.second : c->defaults;
            return RT_OK;
        }
    set_error("Specified scene not found ... (%s)", name.c_str());
    return RT_E_INVALID;
}

int32_t rth_image_height(const rth_user_config *c) {
    return pm::f2i_sat((float)c->image_width / c->aspect_ratio);  // main.rs:1180
}

/* ------------------------------------------------------------------ camera */

void rth_camera_new(const float look_from[3], const float look_at[3], const float world_up[3], float vfov_degrees,
                    float aspect, float aperture, float focus_dist, float time0, float time1, rt_camera *out) {
    V3 from = v3(look_from[0], look_from[1], look_from[2]);
    V3 at = v3(look_at[0], look_at[1], look_at[2]);
    V3 up = v3(world_up[0], world_up[1], world_up[2]);
    float theta = (vfov_degrees * 3.14159265358979323846f) / 180.0f;  // types.rs:44-46
    float half_height = std::tan(theta / 2.0f);
    float half_width = aspect * half_height;
    V3 w = normalize(from - at);
    V3 u = normalize(cross(up, w));
    V3 v = cross(w, u);
    // camera.rs:67-72: `focus_dist * v * half_height` is (f32 * Vec3) * f32
    V3 ll = from - (focus_dist * v) * half_height - (focus_dist * u) * half_width - focus_dist * w;
    V3 hor = ((focus_dist * u) * half_width) * 2.0f;
    V3 ver = ((focus_dist * v) * half_height) * 2.0f;
    auto put = [](float *d, V3 a) { d[0] = a.x; d[1] = a.y; d[2] = a.z; };
    put(out->origin, from);
    put(out->lower_left_corner, ll);
    put(out->horizontal, hor);
    put(out->vertical, ver);
    put(out->u, u);
    put(out->v, v);
    put(out->w, w);
    out->lens_radius = aperture * 0.5f;
    out->time0 = time0;
    out->time1 = time1;
}

/* ------------------------------------------------------------------ sampler */

}  // extern "C"

struct rth_sampler {
    uint32_t sets, spp;
    std::vector<float> samples, disk;
    std::vector<uint32_t> shuffled;
};

extern "C" {

int rth_sampler_new(int32_t spp, int32_t sets, uint32_t seed, rth_sampler **out) {
    if (spp <= 0 || !out) { set_error("sampler: spp must be positive"); return RT_E_INVALID; }
    if (sets <= 0) sets = 83;  // sampling.rs:31
    auto s = std::make_unique<rth_sampler>();
    s->sets = (uint32_t)sets;
    s->spp = (uint32_t)spp;
    size_t total = (size_t)sets * spp;
    s->samples.resize(total * 2);
    rts::Rng rng;
    rts::host_rng(rng, seed, 0x5a3b1e00u);

    // sampling.rs:233-253.  n = floor(sqrt(spp)); the reference emits n*n points per set and
    // then indexes spp per set, which only works for square spp.  Extension for the rest:
    // points n*n..spp-1 sit on the diagonal of the spp x spp sub-cell grid, so that every set
    // still has exactly one point per sub-cell column and row before the shuffles.
    uint32_t n = (uint32_t)std::floor(std::sqrt((double)spp));
    while ((uint64_t)(n + 1) * (n + 1) <= (uint64_t)spp) ++n;
    while ((uint64_t)n * n > (uint64_t)spp) --n;
    float subcell = 1.0f / (float)spp;
    for (int p = 0; p < sets; ++p) {
        float *set = &s->samples[(size_t)p * spp * 2];
        uint32_t k = 0;
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t j = 0; j < n; ++j, ++k) {
                set[k * 2 + 0] = (float)(i * n + j) * subcell + rts::random_real(rng) * subcell;
                set[k * 2 + 1] = (float)(j * n + i) * subcell + rts::random_real(rng) * subcell;
            }
        for (; k < (uint32_t)spp; ++k) {
            set[k * 2 + 0] = (float)k * subcell + rts::random_real(rng) * subcell;
            set[k * 2 + 1] = (float)k * subcell + rts::random_real(rng) * subcell;
        }
    }
    // sampling.rs:36-54: shuffle x, then y
    for (int c = 0; c < 2; ++c)
        for (int p = 0; p < sets; ++p)
            for (int i = 0; i < spp - 1; ++i) {
                size_t target = (size_t)rts::random_int(rng, 0, spp - 1) + (size_t)p * spp;
                size_t src = (size_t)i + (size_t)p * spp + 1;
                float t = s->samples[src * 2 + c];
                s->samples[src * 2 + c] = s->samples[target * 2 + c];
                s->samples[target * 2 + c] = t;
            }
    // sampling.rs:56-65: the SAME index vector is shuffled again for every set
    std::vector<uint32_t> idx(spp);
    for (int i = 0; i < spp; ++i) idx[i] = (uint32_t)i;
    s->shuffled.reserve(total);
    for (int p = 0; p < sets; ++p) {
        for (int i = spp - 1; i >= 1; --i) {
            int j = rts::random_int(rng, 0, i);
            std::swap(idx[i], idx[j]);
        }
        s->shuffled.insert(s->shuffled.end(), idx.begin(), idx.end());
    }
    // sampling.rs:69-107: Shirley concentric map
    s->disk.resize(total * 2);
    for (size_t k = 0; k < total; ++k) {
        float spx = 2.0f * s->samples[k * 2] - 1.0f, spy = 2.0f * s->samples[k * 2 + 1] - 1.0f;
        float r, phi;
        if (spx > -spy) {
            if (spx > spy) { r = spx; phi = spy / spx; }
            else { r = spy; phi = 2.0f - spx / spy; }
        } else {
            if (spx < spy) { r = -spx; phi = 2.0f * 2.0f + spy / spx; }
            else { r = -spy; phi = (spy != 0.0f) ? 6.0f - spx / spy : 0.0f; }
        }
        phi = phi * 0.25f;  // sic: the reference scales by 0.25, not pi/4 (sampling.rs:99)
        float sn, cs;
        pm::sincosf_(phi, &sn, &cs);
        s->disk[k * 2] = r * cs;
        s->disk[k * 2 + 1] = r * sn;
    }
    *out = s.release();
    return RT_OK;
}

void rth_sampler_free(rth_sampler *s) { delete s; }

void rth_sampler_desc(const rth_sampler *s, rt_sampler_desc *out) {
    out->sets = s->sets;
    out->samples_in_set = s->spp;
    out->samples = s->samples.data();
    out->disk_samples = s->disk.data();
    out->shuffled_indices = s->shuffled.data();
}

/* ------------------------------------------------------------------ math hooks */

void rth_mat4_mul(const float a[16], const float b[16], float out[16]) {
    M4 x, y;
    memcpy(x.a, a, 64);
    memcpy(y.a, b, 64);
    M4 r = m4_mul(x, y);
    memcpy(out, r.a, 64);
}

int rth_mat4_invert(const float m[16], float out[16]) {
    M4 x, r;
    memcpy(x.a, m, 64);
    if (!m4_invert(x, &r)) { set_error("Matrix is not invertible"); return RT_E_INVALID; }
    memcpy(out, r.a, 64);
    return RT_OK;
}

void rth_mat4_transpose(const float m[16], float out[16]) {
    M4 x;
    memcpy(x.a, m, 64);
    M4 r = m4_transpose(x);
    memcpy(out, r.a, 64);
}

/* ------------------------------------------------------------------ colours (math/src/colors.rs) */

// NumColorComponent::from_u32 for f32 (colors.rs:34-41): ARGB packed -> r, g, b, a in [0, 1]
void rth_color_from_u32(uint32_t c, float out_rgba[4]) {
    out_rgba[0] = (float)((c >> 16) & 0xFFu) / 255.0f;
    out_rgba[1] = (float)((c >> 8) & 0xFFu) / 255.0f;
    out_rgba[2] = (float)(c & 0xFFu) / 255.0f;
    out_rgba[3] = (float)((c >> 24) & 0xFFu) / 255.0f;
}

// NumColorComponent::to_u32 for u8 (colors.rs:24-26)
uint32_t rth_rgba8_to_u32(uint8_t r, uint8_t g, uint8_t b, uint8_t a) {
    return ((uint32_t)a << 24) | ((uint32_t)r << 16) | ((uint32_t)g << 8) | (uint32_t)b;
}

// From<&str> for RGBAColor (colors.rs:230-262): "rrggbb" or "aarrggbb", optional '#', surrounding blanks trimmed; anything
// else (wrong length, not hexadecimal) is the all-zero colour.  out = r, g, b, a.
void rth_rgba8_from_html(const char *text, uint8_t out_rgba[4]) {
    out_rgba[0] = out_rgba[1] = out_rgba[2] = out_rgba[3] = 0;
    if (!text) return;
    std::string s(text);
    size_t b = 0, e = s.size();
    while (b < e && isspace((unsigned char)s[b])) ++b;
    while (e > b && isspace((unsigned char)s[e - 1])) --e;
    s = s.substr(b, e - b);
    if (s.empty()) return;
    if (s[0] == '#') s = s.substr(1);
    if (!(s.size() == 6 || s.size() == 8)) return;
    uint32_t v = 0;
    for (size_t i = 0; i < s.size(); ++i) {     // u32::from_str_radix(s, 16): digits only (a leading '+' is accepted by Rust: not here)
        const char ch = s[i];
        uint32_t d = ch >= '0' && ch <= '9' ? (uint32_t)(ch - '0') : ch >= 'a' && ch <= 'f' ? (uint32_t)(ch - 'a' + 10)
                     : ch >= 'A' && ch <= 'F' ? (uint32_t)(ch - 'A' + 10) : 16u;
        if (d > 15u) return;
        v = (v << 4) | d;
    }
    if (s.size() == 6) v |= 0xFFu << 24;
    out_rgba[0] = (uint8_t)((v >> 16) & 0xFFu);     // from_u32 for u8 (colors.rs:15-22)
    out_rgba[1] = (uint8_t)((v >> 8) & 0xFFu);
    out_rgba[2] = (uint8_t)(v & 0xFFu);
    out_rgba[3] = (uint8_t)((v >> 24) & 0xFFu);
}

/* ------------------------------------------------------------------ work blocks */

int32_t rth_work_blocks(int32_t width, int32_t height, int32_t block_pixels, int32_t *out, int32_t cap) {
    if (block_pixels <= 0) return 0;
    int32_t bx = width / block_pixels + 1, by = height / block_pixels + 1;  // main.rs:1265-1266
    int32_t n = 0;
    for (int32_t y = 0; y < by; ++y)
        for (int32_t x = 0; x < bx; ++x, ++n) {
            if (!out || n >= cap) continue;
            out[n * 4 + 0] = std::min(x * block_pixels, width);
            out[n * 4 + 1] = std::min((x + 1) * block_pixels, width);
            out[n * 4 + 2] = std::min(y * block_pixels, height);
            out[n * 4 + 3] = std::min((y + 1) * block_pixels, height);
        }
    return n;
}

}  // extern "C"
