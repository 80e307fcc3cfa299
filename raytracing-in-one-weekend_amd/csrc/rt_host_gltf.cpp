// rt_host_gltf.cpp — GLB (binary glTF 2.0) geometry import.
//
// Restates ImportedGeometry::import_from_file / process_nodes / process_node
// (raytracer/src/geometry_import.rs:306-507).  The reference leans on the `gltf` crate
// (1.2.0, not vendored); this file reads the container itself: 12-byte header, JSON chunk,
// BIN chunk.  Kept on purpose: a TRS rotation [x,y,z,w] is read as [w,x,y,z]
// (geometry_import.rs:325 + math/src/quat.rs:120-131) and a node `matrix` is copied
// row-by-row although glTF stores it column-major (geometry_import.rs:316 + mat4.rs:279-313).
// The PBR side (process_materials, geometry_import.rs:172-304; vertex colours, tangents and material ids, :385-425; the
// embedded images as RGBA8, :459-486) is imported too, although TriangleMesh shades with the single material it is given
// (triangle_mesh.rs:238-239) and nothing of it reaches an image.
#include <algorithm>
#include <fstream>
#include <map>

#include "rt_host_internal.h"

using namespace rth;

namespace {

struct J {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<J> arr;
    std::vector<std::pair<std::string, J>> obj;
    const J *get(const char *k) const {
        for (auto &f : obj)
            if (f.first == k) return &f.second;
        return nullptr;
    }
    int as_int(int def = -1) const { return kind == Num ? (int)num : def; }
};

struct JsonParser {
    const char *p, *end;
    bool ok = true;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    bool lit(const char *s) {
        size_t n = strlen(s);
        if ((size_t)(end - p) >= n && memcmp(p, s, n) == 0) { p += n; return true; }
        return false;
    }
    J value() {
        J v;
        ws();
        if (p >= end) { ok = false; return v; }
        if (*p == '{') {
            ++p; v.kind = J::Obj;
            ws();
            if (p < end && *p == '}') { ++p; return v; }
            while (ok) {
                ws();
                J k = value();
                if (k.kind != J::Str) { ok = false; break; }
                ws();
                if (p >= end || *p != ':') { ok = false; break; }
                ++p;
                J val = value();
                v.obj.emplace_back(k.str, std::move(val));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                ok = false;
            }
        } else if (*p == '[') {
            ++p; v.kind = J::Arr;
            ws();
            if (p < end && *p == ']') { ++p; return v; }
            while (ok) {
                v.arr.push_back(value());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                ok = false;
            }
        } else if (*p == '"') {
            ++p; v.kind = J::Str;
            while (p < end && *p != '"') {
                if (*p == '\\' && p + 1 < end) {
                    ++p;
                    switch (*p) {
                    case 'n': v.str += '\n'; break;
                    case 't': v.str += '\t'; break;
                    case 'u': v.str += '?'; p += (end - p > 4) ? 4 : 0; break;
                    default: v.str += *p;
                    }
                    ++p;
                } else {
                    v.str += *p++;
                }
            }
            if (p < end) ++p; else ok = false;
        } else if (lit("true")) { v.kind = J::Bool; v.b = true; }
        else if (lit("false")) { v.kind = J::Bool; }
        else if (lit("null")) { v.kind = J::Null; }
        else {
            char *e = nullptr;
            v.kind = J::Num;
            v.num = strtod(p, &e);
            if (e == p) ok = false;
            p = e;
        }
        return v;
    }
};

struct Accessor {
    const uint8_t *base = nullptr;
    size_t stride = 0, count = 0;
    int component = 0, ncomp = 0;
};

struct Importer {
    const J &doc;
    const uint8_t *bin;
    size_t bin_len;
    Mesh &out;
    rth_geometry &geo;
    std::string err;

    struct NodeRec { int parent; M4 transform; };
    std::vector<NodeRec> recs;

    bool fail(const std::string &m) { if (err.empty()) err = m; return false; }

    bool accessor(int idx, Accessor *a) {
        const J *accs = doc.get("accessors"), *views = doc.get("bufferViews");
        if (!accs || !views || idx < 0 || (size_t)idx >= accs->arr.size()) return fail("accessor out of range");
        const J &ac = accs->arr[idx];
        const J *bv = ac.get("bufferView");
        if (!bv || (size_t)bv->as_int() >= views->arr.size()) return fail("accessor without bufferView (sparse accessors are not supported)");
        const J &view = views->arr[bv->as_int()];
        if (view.get("buffer") && view.get("buffer")->as_int(0) != 0) return fail("only the GLB-embedded buffer 0 is supported");
        size_t off = (size_t)(view.get("byteOffset") ? view.get("byteOffset")->num : 0) + (size_t)(ac.get("byteOffset") ? ac.get("byteOffset")->num : 0);
        a->component = ac.get("componentType") ? ac.get("componentType")->as_int() : 0;
        a->count = ac.get("count") ? (size_t)ac.get("count")->num : 0;
        std::string type = ac.get("type") ? ac.get("type")->str : "";
        a->ncomp = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
        size_t csize = a->component == 5126 || a->component == 5125 ? 4 : a->component == 5123 || a->component == 5122 ? 2 : 1;
        size_t elem = csize * a->ncomp;
        a->stride = view.get("byteStride") ? (size_t)view.get("byteStride")->num : elem;
        if (!a->ncomp || a->stride < elem) return fail("bad accessor type");
        if (a->count && off + (a->count - 1) * a->stride + elem > bin_len) return fail("accessor runs past the BIN chunk");
        a->base = bin + off;
        return true;
    }

    static M4 node_local(const J &node) {
        if (const J *m = node.get("matrix")) {
            if (m->arr.size() == 16) {
                // gltf hands out [[f32;4];4] = columns; `.into()` copies arr[r][c] verbatim.
                M4 r{};
                for (int i = 0; i < 4; ++i)
                    for (int k = 0; k < 4; ++k) r.at(i, k) = (float)m->arr[i * 4 + k].num;
                return r;
            }
        }
        float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
        auto rd = [](const J *a, float *d, size_t n) {
            if (a && a->arr.size() == n) for (size_t i = 0; i < n; ++i) d[i] = (float)a->arr[i].num;
        };
        rd(node.get("translation"), t, 3);
        rd(node.get("rotation"), q, 4);
        rd(node.get("scale"), s, 3);
        M4 sm = m4_scale(v3(s[0], s[1], s[2]));
        M4 rm = quat_to_rotation_matrix(q[0], q[1], q[2], q[3]);  // sic: (w,x,y,z) := glTF (x,y,z,w)
        M4 tm = m4_translate(v3(t[0], t[1], t[2]));
        return m4_mul(m4_mul(tm, rm), sm);
    }

    bool process_node(int idx, int parent) {
        const J *nodes = doc.get("nodes");
        if (!nodes || idx < 0 || (size_t)idx >= nodes->arr.size()) return fail("node index out of range");
        if (recs.size() > 100000) return fail("node graph too large (cycle?)");
        const J &node = nodes->arr[idx];
        M4 local = node_local(node);
        int node_id = (int)recs.size();
        recs.push_back(NodeRec{parent, local});
        if (const J *ch = node.get("children"))
            for (const J &c : ch->arr)
                if (!process_node(c.as_int(), node_id)) return false;
        const J *mesh_ix = node.get("mesh");
        if (!mesh_ix) return true;
        M4 matrix = local;
        for (int p = parent; p >= 0; p = recs[p].parent) matrix = m4_mul(recs[p].transform, matrix);
        recs[node_id].transform = matrix;
        M4 inv;
        if (!m4_invert(matrix, &inv)) return fail("Matrix is not invertible");
        M4 nm = m4_transpose(inv);

        const J *meshes = doc.get("meshes");
        if (!meshes || (size_t)mesh_ix->as_int() >= meshes->arr.size()) return fail("mesh index out of range");
        const J *prims = meshes->arr[mesh_ix->as_int()].get("primitives");
        size_t idx_start = out.indices.size();
        for (const J &prim : prims ? prims->arr : std::vector<J>{}) {
            size_t vertex_start = out.verts.size();
            // geometry_import.rs:363-375: a named material is required
            const J *mats = doc.get("materials"), *mi = prim.get("material");
            if (!mats || !mi || (size_t)mi->as_int() >= mats->arr.size() || !mats->arr[mi->as_int()].get("name"))
                return fail("Materials without names are not supported chief ...");
            // ... and it is found by NAME in the table process_materials made (:368-375); two materials of one name cannot exist (:268)
            int material_index = -1;
            {
                const std::string &nm_ = mats->arr[mi->as_int()].get("name")->str;
                for (size_t k = 0; k < geo.pbr_names.size(); ++k) if (geo.pbr_names[k] == nm_) { material_index = (int)k; break; }
            }
            if (material_index < 0) return fail("Fatal error: material not found in PBR materials buffer.");
            const J *attrs = prim.get("attributes");
            const J *pos = attrs ? attrs->get("POSITION") : nullptr;
            if (!pos) return fail("Missing positions attribute on mesh");
            Accessor a;
            if (!accessor(pos->as_int(), &a)) return false;
            if (a.component != 5126 || a.ncomp != 3) return fail("POSITION must be float VEC3");
            for (size_t i = 0; i < a.count; ++i) {
                float v[3];
                memcpy(v, a.base + i * a.stride, 12);
                // Mat4 * Vec4(x,y,z,1), mat4.rs:434-441
                rtg_vertex o{};
                o.pos[0] = matrix.at(0, 0) * v[0] + matrix.at(0, 1) * v[1] + matrix.at(0, 2) * v[2] + matrix.at(0, 3) * 1.0f;
                o.pos[1] = matrix.at(1, 0) * v[0] + matrix.at(1, 1) * v[1] + matrix.at(1, 2) * v[2] + matrix.at(1, 3) * 1.0f;
                o.pos[2] = matrix.at(2, 0) * v[0] + matrix.at(2, 1) * v[1] + matrix.at(2, 2) * v[2] + matrix.at(2, 3) * 1.0f;
                out.verts.push_back(o);
                rth_vertex_attrs x{};      // GeometryVertex::default (geometry_import.rs:57-68) + the primitive's material
                x.color[3] = 1.0f;
                x.pbr_buf_id = (uint32_t)material_index;
                geo.vertex_attrs.push_back(x);
            }
            if (const J *nr = attrs->get("NORMAL")) {
                if (!accessor(nr->as_int(), &a)) return false;
                if (a.component != 5126 || a.ncomp != 3) return fail("NORMAL must be float VEC3");
                for (size_t i = 0; i < a.count && vertex_start + i < out.verts.size(); ++i) {
                    float v[3];
                    memcpy(v, a.base + i * a.stride, 12);
                    V3 n = v3(nm.at(0, 0) * v[0] + nm.at(0, 1) * v[1] + nm.at(0, 2) * v[2] + nm.at(0, 3) * 0.0f,
                              nm.at(1, 0) * v[0] + nm.at(1, 1) * v[1] + nm.at(1, 2) * v[2] + nm.at(1, 3) * 0.0f,
                              nm.at(2, 0) * v[0] + nm.at(2, 1) * v[1] + nm.at(2, 2) * v[2] + nm.at(2, 3) * 0.0f);
                    n = normalize(n);
                    float *d = out.verts[vertex_start + i].normal;
                    d[0] = n.x; d[1] = n.y; d[2] = n.z;
                }
            }
            if (const J *uv = attrs->get("TEXCOORD_0")) {
                if (!accessor(uv->as_int(), &a)) return false;
                if (a.ncomp != 2) return fail("TEXCOORD_0 must be VEC2");
                for (size_t i = 0; i < a.count && vertex_start + i < out.verts.size(); ++i) {
                    float *d = out.verts[vertex_start + i].uv;
                    const uint8_t *s = a.base + i * a.stride;
                    if (a.component == 5126) memcpy(d, s, 8);
                    else if (a.component == 5121) { d[0] = (float)s[0] / 255.0f; d[1] = (float)s[1] / 255.0f; }
                    else if (a.component == 5123) { uint16_t h[2]; memcpy(h, s, 4); d[0] = (float)h[0] / 65535.0f; d[1] = (float)h[1] / 65535.0f; }
                    else return fail("TEXCOORD_0 component type not supported");
                }
            }
            if (const J *tg = attrs->get("TANGENT")) {      // read_tangents: float VEC4 (:415-419)
                if (!accessor(tg->as_int(), &a)) return false;
                if (a.component != 5126 || a.ncomp != 4) return fail("TANGENT must be float VEC4");
                for (size_t i = 0; i < a.count && vertex_start + i < out.verts.size(); ++i) memcpy(geo.vertex_attrs[vertex_start + i].tangent, a.base + i * a.stride, 16);
            }
            if (const J *cl = attrs->get("COLOR_0")) {      // read_colors(0).into_rgba_f32 (:421-425): RGB gets alpha 1, integers are normalised
                if (!accessor(cl->as_int(), &a)) return false;
                if (a.ncomp != 3 && a.ncomp != 4) return fail("COLOR_0 must be VEC3 or VEC4");
                for (size_t i = 0; i < a.count && vertex_start + i < out.verts.size(); ++i) {
                    float *d = geo.vertex_attrs[vertex_start + i].color;
                    const uint8_t *sp = a.base + i * a.stride;
                    d[3] = 1.0f;
                    for (int c = 0; c < a.ncomp; ++c) {
                        if (a.component == 5126) memcpy(&d[c], sp + 4 * c, 4);
                        else if (a.component == 5121) d[c] = (float)sp[c] / 255.0f;
                        else if (a.component == 5123) { uint16_t h; memcpy(&h, sp + 2 * c, 2); d[c] = (float)h / 65535.0f; }
                        else return fail("COLOR_0 component type not supported");
                    }
                }
            }
            const J *ind = prim.get("indices");
            if (!ind) return fail("Missing indices on mesh");
            if (!accessor(ind->as_int(), &a)) return false;
            for (size_t i = 0; i < a.count; ++i) {
                const uint8_t *s = a.base + i * a.stride;
                uint32_t v;
                if (a.component == 5125) memcpy(&v, s, 4);
                else if (a.component == 5123) { uint16_t h; memcpy(&h, s, 2); v = h; }
                else if (a.component == 5121) v = *s;
                else return fail("index component type not supported");
                out.indices.push_back(v + (uint32_t)vertex_start);
            }
        }
        out.nodes.push_back(rtg_mesh_node{(uint32_t)idx_start, (uint32_t)(out.indices.size() - idx_start)});
        return true;
    }

    // texture index -> image index (`.texture().source().index()`, geometry_import.rs:184-186)
    bool texture_source(const J *tex_info, uint32_t *image) {
        const J *textures = doc.get("textures"), *ti = tex_info ? tex_info->get("index") : nullptr;
        if (!textures || !ti || ti->as_int() < 0 || (size_t)ti->as_int() >= textures->arr.size()) return fail("texture index out of range");
        const J *src = textures->arr[ti->as_int()].get("source");
        if (!src || src->as_int() < 0) return fail("texture without an image source");
        *image = (uint32_t)src->as_int();
        return true;
    }

    // process_materials (geometry_import.rs:172-304).  Every material needs a name, a base-colour, a metallic-roughness and a
    // normal texture, otherwise the reference panics (:176-201).  Per kind the distinct source images, ascending, are the layers
    // of a texture array (:225-263); a material stores the layer, not the image (:270-297).
    bool process_materials() {
        const J *mats = doc.get("materials");
        struct Def { std::string name; uint32_t src[3]; float base[4], metallic, roughness; };
        std::vector<Def> defs;
        for (const J &m : mats ? mats->arr : std::vector<J>{}) {
            const J *pbr = m.get("pbrMetallicRoughness");
            if (!m.get("name") || m.get("name")->kind != J::Str) return fail("Unnamed materials are not supported!");
            if (!pbr || !pbr->get("baseColorTexture") || !pbr->get("metallicRoughnessTexture") || !m.get("normalTexture"))
                return fail("material lacks base colour / metallic-roughness / normal texture");
            Def d{};
            d.name = m.get("name")->str;
            if (!texture_source(pbr->get("baseColorTexture"), &d.src[0]) || !texture_source(pbr->get("metallicRoughnessTexture"), &d.src[1]) ||
                !texture_source(m.get("normalTexture"), &d.src[2])) return false;
            d.base[0] = d.base[1] = d.base[2] = d.base[3] = 1.0f;      // glTF defaults: factor 1,1,1,1, metallic 1, roughness 1
            if (const J *f = pbr->get("baseColorFactor")) if (f->arr.size() == 4) for (int i = 0; i < 4; ++i) d.base[i] = (float)f->arr[i].num;
            d.metallic = pbr->get("metallicFactor") ? (float)pbr->get("metallicFactor")->num : 1.0f;
            d.roughness = pbr->get("roughnessFactor") ? (float)pbr->get("roughnessFactor")->num : 1.0f;
            defs.push_back(d);
        }
        for (int kind = 0; kind < 3; ++kind) {
            std::vector<uint32_t> &l = geo.image_list[kind];
            for (const Def &d : defs) l.push_back(d.src[kind]);
            std::sort(l.begin(), l.end());
            l.erase(std::unique(l.begin(), l.end()), l.end());
        }
        auto layer_of = [&](int kind, uint32_t image) {
            const std::vector<uint32_t> &l = geo.image_list[kind];
            return (uint32_t)(std::lower_bound(l.begin(), l.end(), image) - l.begin());
        };
        for (const Def &d : defs) {
            for (const std::string &n : geo.pbr_names) if (n == d.name) return fail("two materials of one name");   // the assert of :268
            rth_pbr_material m{};
            m.base_color_factor[0] = d.base[0]; m.base_color_factor[1] = d.base[1]; m.base_color_factor[2] = d.base[2];
            m.metallic_factor = d.metallic; m.roughness_factor = d.roughness;
            m.base_color_texarray_id = layer_of(0, d.src[0]);
            m.metallic_rough_texarray_id = layer_of(1, d.src[1]);
            m.normal_texarray_id = layer_of(2, d.src[2]);
            geo.pbr_materials.push_back(m);
            geo.pbr_names.push_back(d.name);
        }
        return true;
    }

    // gltf::import_slice decodes every image of the document; the reference then widens RGB8 to RGBA8 (geometry_import.rs:459-486).
    // Here: the host library's own decoder, which gives RGBA8 for every format it reads (into_rgba8).  An image it cannot read (or
    // one that lives in an external file: a .glb embeds its images) stays empty and fails only if a material list asks for it.
    void decode_images() {
        const J *imgs = doc.get("images"), *views = doc.get("bufferViews");
        geo.images.resize(imgs ? imgs->arr.size() : 0);
        for (size_t i = 0; i < geo.images.size(); ++i) {
            const J *bv = imgs->arr[i].get("bufferView");
            if (!bv || !views || bv->as_int() < 0 || (size_t)bv->as_int() >= views->arr.size()) continue;
            const J &view = views->arr[bv->as_int()];
            // offsets and lengths are JSON numbers: anything that is not a whole number inside the BIN chunk (negative, NaN, 1e300 —
            // a cast of those is undefined, and a sum that wraps would pass the range check) leaves the image empty
            const double off_d = view.get("byteOffset") ? view.get("byteOffset")->num : 0.0, len_d = view.get("byteLength") ? view.get("byteLength")->num : 0.0;
            if (!(off_d >= 0.0 && off_d <= (double)bin_len) || !(len_d >= 1.0 && len_d <= (double)bin_len)) continue;
            const size_t off = (size_t)off_d, len = (size_t)len_d;
            if ((view.get("buffer") && view.get("buffer")->as_int(0) != 0) || len > bin_len || off > bin_len - len) continue;
            int32_t w = 0, h = 0;
            uint8_t *px = nullptr;
            if (rth_image_decode(bin + off, len, &w, &h, &px) != RT_OK) continue;
            geo.images[i].width = (uint32_t)w; geo.images[i].height = (uint32_t)h; geo.images[i].rgba = px;
        }
        for (int kind = 0; kind < 3; ++kind)
            for (uint32_t image : geo.image_list[kind]) {
                const bool have = image < geo.images.size() && geo.images[image].rgba;
                geo.layer_ptrs[kind].push_back(have ? geo.images[image].rgba : nullptr);
                geo.layer_sizes[kind].push_back(have ? geo.images[image].width : 0u);
                geo.layer_sizes[kind].push_back(have ? geo.images[image].height : 0u);
            }
    }

    bool run() {
        if (!process_materials()) return false;
        decode_images();
        const J *scenes = doc.get("scenes");
        for (const J &sc : scenes ? scenes->arr : std::vector<J>{})
            if (const J *ns = sc.get("nodes"))
                for (const J &n : ns->arr)
                    if (!process_node(n.as_int(), -1)) return false;
        return true;
    }
};

}  // namespace

extern "C" {

int rth_geometry_import_glb(const char *path, rth_geometry **out) {
    if (!path || !out) { set_error("null argument"); return RT_E_INVALID; }
    std::ifstream f(path, std::ios::binary);
    if (!f) { set_error("Failed to open geometry file %s", path); return RT_E_INVALID; }
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    auto rd32 = [&](size_t o) { uint32_t v; memcpy(&v, &data[o], 4); return v; };
    if (data.size() < 20 || rd32(0) != 0x46546c67u || rd32(4) != 2u) { set_error("GLTF import error: not a GLB v2 file"); return RT_E_INVALID; }
    size_t total = std::min<size_t>(rd32(8), data.size());
    size_t jlen = rd32(12);
    if (rd32(16) != 0x4e4f534au || 20 + jlen > total) { set_error("GLTF import error: first chunk is not JSON"); return RT_E_INVALID; }
    const uint8_t *bin = nullptr;
    size_t bin_len = 0;
    size_t o = 20 + ((jlen + 3) & ~size_t(3));
    if (o + 8 <= total && rd32(o + 4) == 0x004e4942u) {
        bin_len = rd32(o);
        bin = &data[o + 8];
        if (o + 8 + bin_len > total) { set_error("GLTF import error: BIN chunk truncated"); return RT_E_INVALID; }
    }
    JsonParser jp{(const char *)&data[20], (const char *)&data[20] + jlen};
    J doc = jp.value();
    if (!jp.ok || doc.kind != J::Obj) { set_error("GLTF import error: bad JSON"); return RT_E_INVALID; }
    auto geo = std::make_unique<rth_geometry>();
    Importer imp{doc, bin, bin_len, geo->mesh, *geo, {}, {}};
    if (!imp.run()) { set_error("GLTF import error: %s", imp.err.c_str()); return RT_E_INVALID; }
    *out = geo.release();
    return RT_OK;
}

void rth_geometry_free(rth_geometry *g) { delete g; }

void rth_geometry_view(const rth_geometry *g, const rtg_vertex **verts, uint32_t *n_verts, const uint32_t **indices,
                       uint32_t *n_indices, const rtg_mesh_node **nodes, uint32_t *n_nodes) {
    *verts = g->mesh.verts.data();     *n_verts = (uint32_t)g->mesh.verts.size();
    *indices = g->mesh.indices.data(); *n_indices = (uint32_t)g->mesh.indices.size();
    *nodes = g->mesh.nodes.data();     *n_nodes = (uint32_t)g->mesh.nodes.size();
}

void rth_geometry_pbr_materials(const rth_geometry *g, const rth_pbr_material **materials, uint32_t *n_materials) {
    *materials = g->pbr_materials.data(); *n_materials = (uint32_t)g->pbr_materials.size();
}

void rth_geometry_vertex_attrs(const rth_geometry *g, const rth_vertex_attrs **attrs, uint32_t *n_verts) {
    *attrs = g->vertex_attrs.data(); *n_verts = (uint32_t)g->vertex_attrs.size();
}

int32_t rth_geometry_material_index(const rth_geometry *g, const char *name) {
    if (!g || !name) return -1;
    for (size_t k = 0; k < g->pbr_names.size(); ++k) if (g->pbr_names[k] == name) return (int32_t)k;
    return -1;
}

int rth_geometry_pbr_images(const rth_geometry *g, int kind, uint32_t *width, uint32_t *height, uint32_t *n_layers,
                            const uint8_t *const **rgba_layers, const uint32_t **layer_sizes) {
    if (!g || kind < 0 || kind > 2 || !width || !height || !n_layers || !rgba_layers) { set_error("null argument or kind outside 0..2"); return RT_E_INVALID; }
    const int size_kind = kind == 2 ? 1 : kind;     // sic: pbr_normal_images takes its size from the metallic-roughness list (geometry_import.rs:135)
    if (g->layer_ptrs[kind].empty() || g->layer_ptrs[size_kind].empty()) { set_error("the file has no materials"); return RT_E_INVALID; }   // images[..[0]] panics
    for (const uint8_t *p : g->layer_ptrs[kind])
        if (!p) { set_error("an image of this list is not embedded in the file or could not be decoded"); return RT_E_UNSUPPORTED_PRIMITIVE; }
    if (!g->layer_ptrs[size_kind][0]) { set_error("the first metallic-roughness image could not be decoded"); return RT_E_UNSUPPORTED_PRIMITIVE; }
    *width = g->layer_sizes[size_kind][0]; *height = g->layer_sizes[size_kind][1];
    *n_layers = (uint32_t)g->layer_ptrs[kind].size();
    *rgba_layers = g->layer_ptrs[kind].data();
    if (layer_sizes) *layer_sizes = g->layer_sizes[kind].data();
    return RT_OK;
}

}  // extern "C"
