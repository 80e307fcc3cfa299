// rt_host_internal.h — shared pieces of the host library (not installed).
#ifndef RT_HOST_INTERNAL_H
#define RT_HOST_INTERNAL_H

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "rt_host.h"
#include "rt_portable_math.h"
#include "rt_seed_schedule.h"

namespace rth {

void set_error(const char *fmt, ...);

struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline float f32_min(float a, float b) { return a != a ? b : (b != b ? a : (a < b ? a : b)); }  // f32::min
inline float f32_max(float a, float b) { return a != a ? b : (b != b ? a : (a > b ? a : b)); }  // f32::max
// math/src/vec3.rs:690-700
inline V3 normalize(V3 v) {
    float lensq = length_squared(v);
    if (lensq == 0.0f) return {0.0f, 0.0f, 0.0f};
    float inv = 1.0f / std::sqrt(lensq);
    return v * inv;
}

// Row-major 4x4, a[r*4+c] (math/src/mat4.rs:20-41).
struct M4 {
    float a[16];
    float &at(int r, int c) { return a[r * 4 + c]; }
    float at(int r, int c) const { return a[r * 4 + c]; }
};
M4 m4_identity();
M4 m4_mul(const M4 &l, const M4 &r);          // mat4.rs:512-531
M4 m4_transpose(const M4 &m);                 // mat4.rs:52-74
float m4_det(const M4 &m);                    // mat4.rs:536-561 (typos kept)
M4 m4_adjoint(const M4 &m);                   // mat4.rs:567-606
bool m4_invert(const M4 &m, M4 *out);         // mat4.rs:608-617
V3 m4_transform_point(const M4 &m, V3 p);     // mat4.rs:444-453
V3 m4_transform_vector(const M4 &m, V3 p);    // mat4.rs:455-464
M4 m4_translate(V3 t);                        // mat4.rs:76-101
M4 m4_scale(V3 s);                            // mat4.rs:103-125
M4 quat_to_rotation_matrix(float w, float x, float y, float z);     // quat.rs:266-298
void quat_axis_angle(float angle_deg, V3 axis, float q[4] /*w,x,y,z*/);  // quat.rs:66-88
// Hyperboloid::new (hyperboloid.rs:26-67): out = p1 xyz, p2 xyz (after the swap), zmin, zmax, rmax, ah, ch
bool hyperboloid_derive(const float p1[3], const float p2[3], float out[11]);

struct Mesh {
    std::vector<rtg_vertex> verts;
    std::vector<uint32_t> indices;
    std::vector<rtg_mesh_node> nodes;
};

}  // namespace rth

struct rth_geometry {
    rth::Mesh mesh;  // imported (object-space) vertices
    // the PBR side (geometry_import.rs:172-304): material table, per-vertex extras, the three image lists
    std::vector<rth_pbr_material> pbr_materials;
    std::vector<std::string> pbr_names;              // gltf_mat_2_pbr_mat_mapping, in table order
    std::vector<rth_vertex_attrs> vertex_attrs;
    struct Image { uint32_t width = 0, height = 0; uint8_t *rgba = nullptr; };
    std::vector<Image> images;                       // every image of the document, decoded to RGBA8 (null if it could not be)
    std::vector<uint32_t> image_list[3];             // per kind: the distinct source image indices, ascending
    std::vector<const uint8_t *> layer_ptrs[3];
    std::vector<uint32_t> layer_sizes[3];
    ~rth_geometry() { for (Image &im : images) if (im.rgba) rth_image_free(im.rgba); }
};

struct rth_scene {
    std::vector<rtg_node> nodes;
    std::vector<int32_t> children;
    std::vector<rtg_material> materials;
    std::vector<rtg_texture> textures;
    std::vector<rtg_perlin> perlins;
    std::vector<std::vector<uint8_t>> image_pixels;
    std::vector<rtg_image> images;
    std::vector<std::unique_ptr<rth::Mesh>> meshes;
    std::vector<rtg_mesh> mesh_views;
    int32_t world_root = -1, lights_root = -1;
    rtg_graph view{};
};

#endif
