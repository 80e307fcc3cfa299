/*
 * rt_host.h — C-ABI of the host side (librtiow_host.so): everything the reference does
 * BEFORE the worker pool starts, restated so that it feeds the HIP backend.
 *
 *   RaytracerState::load_config + RaytracerParams::from   raytracer/src/main.rs:1178-1263
 *   Camera::new                                            raytracer/src/camera.rs:39-80
 *   MultiJitteredSampler::new                              raytracer/src/sampling.rs:30-135, 233-253
 *   scene_random_world / scene_cornell_box /
 *   scene_final_chapter2 / scene_mesh                      raytracer/src/main.rs:106-199, 362-522, 685-828, 969-1029
 *   scene_geometric_primitives / scene_blocks              raytracer/src/main.rs:830-967, 1031-1133
 *   scene_two_spheres / _two_perlin_spheres /
 *   _textured_spheres / _simple_light                      raytracer/src/main.rs:201-360
 *   ImportedGeometry (GLB) + TriangleMesh::new             raytracer/src/geometry_import.rs:306-507, triangle_mesh.rs:43-119
 *   "Hittable::flatten" (does not exist in the reference)  object tree -> flat arrays + BVH of include/rt_api.h
 *
 * No GPU code here; it builds and runs without HIP.  All functions return 0 on success and a
 * negative rt_status otherwise; rth_last_error() gives the text (thread-local).
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "rt_api.h"
#include "rt_scene_graph.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *rth_last_error(void);

/* ------------------------------------------------------------------ scene graph building */
typedef struct rth_scene rth_scene;

rth_scene *rth_scene_new(void);
void       rth_scene_free(rth_scene *s);
/* The graph view stays valid until the next mutation of the scene. */
const rtg_graph *rth_scene_graph(rth_scene *s);

/* Textures / materials: constructors of the reference types; return the new index (>=0). */
int rth_add_solid_texture(rth_scene *s, float r, float g, float b);               /* solid_color_texture.rs:9-13 */
int rth_add_checker_texture(rth_scene *s, int odd_tex, int even_tex, float repeat); /* checker_texture.rs:23-33 */
int rth_add_noise_texture(rth_scene *s, float scale, uint32_t seed);              /* noise_texture.rs:13-18 + perlin.rs:65-82 */
int rth_add_noise_texture_table(rth_scene *s, float scale, const rtg_perlin *table); /* same, caller-made PerlinNoise tables */
int rth_add_image_texture(rth_scene *s, int width, int height, const uint8_t *rgba); /* image_texture.rs:40-47; 0x0 = blank() */
int rth_add_lambertian(rth_scene *s, int albedo_tex);                             /* lambertian.rs:27-29 */
int rth_add_metal(rth_scene *s, float r, float g, float b, float fuzz);           /* metal.rs:12-17 */
int rth_add_dielectric(rth_scene *s, float ior);                                  /* dielectric.rs:11-13 */
int rth_add_diffuse_light(rth_scene *s, int emit_tex);                            /* diffuse_light.rs:18-20 */
int rth_add_isotropic(rth_scene *s, int albedo_tex);                              /* isotropic.rs:10-12 */

/* Hittables; children are node indices.  Return the new node index. */
int rth_add_sphere(rth_scene *s, const float c[3], float r, int mtl);
int rth_add_moving_sphere(rth_scene *s, const float c0[3], const float c1[3], float t0, float t1, float r, int mtl);
int rth_add_rect(rth_scene *s, int rtg_type /* RTG_XY_RECT.. */, float a0, float a1, float b0, float b1, float k, int mtl);
int rth_add_block(rth_scene *s, const float p0[3], const float p1[3], int mtl);
int rth_add_flip_face(rth_scene *s, int child);
int rth_add_translate(rth_scene *s, int child, const float offset[3]);
int rth_add_rotate_y(rth_scene *s, int child, float degrees);
/* RotateY as the reference holds it after RotateY::new (transform.rs:57-62, 65-67): the host's own sin and cos */
int rth_add_rotate_y_sincos(rth_scene *s, int child, float sin_theta, float cos_theta);
int rth_add_bvh(rth_scene *s, const int32_t *children, int n);
int rth_add_list(rth_scene *s, const int32_t *children, int n);
int rth_add_constant_medium(rth_scene *s, int boundary, int albedo_tex, float density);
/* TriangleMesh::new: vertices are IMPORTED (object-space) vertices, obj2world row-major 4x4
 * (math/src/mat4.rs:20-41); the call bakes them to world space as triangle_mesh.rs:48-72.
 * obj2world = NULL: the vertices are already baked (TriangleMesh::vertices as the reference keeps them) and are taken as is. */
int rth_add_triangle_mesh(rth_scene *s, const rtg_vertex *verts, uint32_t n_verts, const uint32_t *indices,
                          uint32_t n_indices, const rtg_mesh_node *nodes, uint32_t n_nodes,
                          const float obj2world[16], int mtl);
/* Transform::new (transform.rs:162-175): obj2world row-major 4x4 whose last row must be 0,0,0,1 (every Transform
 * the reference builds is affine); RT_E_INVALID when it cannot be inverted (the reference asserts, mat4.rs:610). */
int rth_add_transform(rth_scene *s, int child, const float obj2world[16]);
/* Quadrics in their own frame, axis = z: Cylinder::new (cylinder.rs:32-44), Cone::new (cone.rs:20-27),
 * Paraboloid::new (paraboloid.rs:23-37), Hyperboloid::new (hyperboloid.rs:26-67).  phi_max in radians. */
int rth_add_cylinder(rth_scene *s, float radius, float z0, float z1, float phi_max, int mtl);
int rth_add_cone(rth_scene *s, float radius, float height, float phi_max, int mtl);
int rth_add_paraboloid(rth_scene *s, float radius, float zmin, float zmax, float phi_max, int mtl);
int rth_add_hyperboloid(rth_scene *s, const float p1[3], const float p2[3], float phi_max, int mtl);
int rth_set_roots(rth_scene *s, int world_list, int lights_list);

/* ------------------------------------------------------------------ glTF import */
typedef struct rth_geometry rth_geometry;
/* ImportedGeometry::import_from_file for a .glb (positions, normals, uvs, indices, nodes; children
 * before the node's own mesh; the importer's quaternion/matrix quirks of geometry_import.rs:314-329
 * are reproduced; materials, colours, tangents and images: see below). */
int  rth_geometry_import_glb(const char *path, rth_geometry **out);
void rth_geometry_free(rth_geometry *g);
void rth_geometry_view(const rth_geometry *g, const rtg_vertex **verts, uint32_t *n_verts, const uint32_t **indices,
                       uint32_t *n_indices, const rtg_mesh_node **nodes, uint32_t *n_nodes);

/* The PBR side of the import (geometry_import.rs:15-24, 47-55, 172-304).  The reference fills these tables and
 * its renderer then never looks at them (triangle_mesh.rs:181, 238-239: the mesh is shaded with the one material it was given),
 * so they cannot change an image; they are here so that a host that switches to this library finds what ImportedGeometry held.
 * PbrMaterial, field for field (repr(C, align(16)): 32 bytes): */
typedef struct rth_pbr_material {
    float    base_color_factor[3];          /* baseColorFactor.xyz (alpha dropped, geometry_import.rs:291) */
    float    metallic_factor, roughness_factor;
    uint32_t base_color_texarray_id;        /* layer in the base-colour image list below, not the glTF image index */
    uint32_t metallic_rough_texarray_id;
    uint32_t normal_texarray_id;
} rth_pbr_material;
/* what GeometryVertex carries besides pos / normal / uv (geometry_import.rs:47-68); parallel to rth_geometry_view's vertices */
typedef struct rth_vertex_attrs {
    float    color[4];      /* COLOR_0 as rgba f32, default 0,0,0,1 */
    float    tangent[4];    /* TANGENT, default 0,0,0,0 */
    uint32_t pbr_buf_id;    /* index into the material table: the primitive's material, by NAME (geometry_import.rs:363-375) */
} rth_vertex_attrs;
void rth_geometry_pbr_materials(const rth_geometry *g, const rth_pbr_material **materials, uint32_t *n_materials);
void rth_geometry_vertex_attrs(const rth_geometry *g, const rth_vertex_attrs **attrs, uint32_t *n_verts);
/* gltf_mat_2_pbr_mat_mapping: material name -> index into the table, -1 if there is none of that name */
int32_t rth_geometry_material_index(const rth_geometry *g, const char *name);
/* pbr_base_color_images / pbr_metallic_roughness_images / pbr_normal_images (geometry_import.rs:114-142): kind 0 / 1 / 2.  The layers
 * are the distinct source images of that kind in ascending glTF image index (sort + dedup, :225-263), decoded to RGBA8 (:465-486;
 * rth_image_decode).  *width, *height are those of the kind's FIRST layer — for kind 2 of the first metallic-roughness layer, as the
 * reference has it (:135).  layer_sizes (may be NULL) receives width, height per layer.  Pointers stay valid until rth_geometry_free. */
int rth_geometry_pbr_images(const rth_geometry *g, int kind, uint32_t *width, uint32_t *height, uint32_t *n_layers,
                            const uint8_t *const **rgba_layers, const uint32_t **layer_sizes);

/* ------------------------------------------------------------------ the reference's scenes */
typedef struct rth_image { int32_t width, height; const uint8_t *rgba; } rth_image;

typedef struct rth_scene_assets {
    rth_image earthmap;          /* data/textures/misc/earthmap.jpg      (Chapter2Final) */
    rth_image uvgrid01;          /* data/textures/uv_grids/ash_uvgrid01.jpg (MeshTest floor) */
    rth_image uvgrid09;          /* data/textures/uv_grids/ash_uvgrid09.jpg (MeshTest teapot) */
    const rth_geometry *teapot;  /* data/models/teapot/teapot.glb        (MeshTest) */
    int32_t mesh_grid;           /* MeshTest: K -> K x K baked copies of the mesh (1 = the reference scene) */
    rth_image uvgrid03;          /* data/textures/uv_grids/ash_uvgrid03.jpg (GeometricPrimitives) */
    rth_image uvgrid08;          /* data/textures/uv_grids/ash_uvgrid08.jpg (GeometricPrimitives) */
    rth_image uvgrid10;          /* data/textures/uv_grids/ash_uvgrid10.jpg (GeometricPrimitives) */
} rth_scene_assets;

/* name = a `Scene` variant of main.rs:93-104: "RandomWorld", "CornellBox", "Chapter2Final",
 * "MeshTest", "GeometricPrimitives", "Blocks", "TwoSpheres", "PerlinSpheres", "TexturedSpheres", "SimpleLight" — all ten — and
 * "CornellBoxSmoke": scene_cornell_box_smoke (main.rs:524-683), which no `Scene` variant reaches; it is the reference's one use of a
 * ConstantMedium over a boundary that is not a sphere.  seed feeds every random_* call of the builder (the reference is unseeded). */
int rth_build_scene(rth_scene *s, const char *name, uint32_t seed, const rth_scene_assets *assets);

/* ------------------------------------------------------------------ config (RON) */
typedef struct rth_user_config {      /* RaytracerUserConfig, main.rs:1135-1154 */
    int32_t workers, worker_block_pixels;
    float   aspect_ratio;
    int32_t image_width, samples_per_pixel, max_ray_depth;
    float   vertical_fov;
    float   look_from[3], look_at[3], world_up[3];
    float   aperture, focus_dist;
    int32_t shuffle_workblocks;
    float   background[3];
    int32_t projection;               /* 0 Perspective 1 Orthographic 2 FishEye 3 SphericalPanoramic */
    float   psi_max, lambda_max;
} rth_user_config;

typedef struct rth_config rth_config;
int  rth_config_load(const char *path, rth_config **out);       /* main.rs:1241-1246 */
int  rth_config_parse(const char *text, rth_config **out);
void rth_config_free(rth_config *c);
const char *rth_config_active_scene(const rth_config *c);
/* main.rs:1251-1261: params of `scene` (NULL = active scene); None -> default_params. */
int  rth_config_scene_params(const rth_config *c, const char *scene, rth_user_config *out);
/* RaytracerParams::from, main.rs:1180: image_height = (image_width / aspect_ratio) as i32 */
int32_t rth_image_height(const rth_user_config *c);

/* ------------------------------------------------------------------ camera + sampler */
void rth_camera_new(const float look_from[3], const float look_at[3], const float world_up[3], float vfov_degrees,
                    float aspect, float aperture, float focus_dist, float time0, float time1, rt_camera *out);

typedef struct rth_sampler rth_sampler;
/* MultiJitteredSampler::new(spp, None) with 83 sets when sets<=0.  Non-square spp (which
 * panics in the reference, sampling.rs:235-251) is extended: see DESIGN.md. */
int  rth_sampler_new(int32_t spp, int32_t sets, uint32_t seed, rth_sampler **out);
void rth_sampler_free(rth_sampler *s);
void rth_sampler_desc(const rth_sampler *s, rt_sampler_desc *out);

/* ------------------------------------------------------------------ flatten */
typedef struct rth_flat rth_flat;
/* Lower the object tree to the arrays of rt_scene_desc and build the traversal BVH. */
int  rth_flatten(const rtg_graph *g, rth_flat **out);
/* flags: RTH_FLATTEN_DEVICE_BVH = build no tree here; the desc then carries one box per slot and asks
 * rt_scene_upload to build the BVH on the GPU (rt_scene_desc.build_bvh_on_device). */
enum { RTH_FLATTEN_DEVICE_BVH = 1 };
int  rth_flatten_opts(const rtg_graph *g, uint32_t flags, rth_flat **out);
void rth_flat_free(rth_flat *f);
void rth_flat_desc(const rth_flat *f, rt_scene_desc *out);

/* math/src/mat4.rs:512-531 and :608-617, exposed for the reference's own known-answer test
 * (math/src/mat4.rs:641-655).  Row-major 4x4.  rth_mat4_invert returns RT_E_INVALID for det == 0. */
void rth_mat4_mul(const float a[16], const float b[16], float out[16]);
int  rth_mat4_invert(const float m[16], float out[16]);
/* math/src/mat4.rs:52-74, for the reference's own known answer (mat4.rs:657-667) */
void rth_mat4_transpose(const float m[16], float out[16]);
/* math/src/colors.rs, which feeds the palette colours of scene_blocks (main.rs:1031-1133, From<u32>): :34-41 ARGB -> f32 rgba,
 * :24-26 u8 components -> ARGB, :230-262 "rrggbb" / "#aarrggbb" -> u8 r, g, b, a (all zero when it does not parse).  The
 * reference's own known answers (colors.rs:589-629) are in tests/test_host.py. */
void     rth_color_from_u32(uint32_t argb, float out_rgba[4]);
uint32_t rth_rgba8_to_u32(uint8_t r, uint8_t g, uint8_t b, uint8_t a);
void     rth_rgba8_from_html(const char *text, uint8_t out_rgba[4]);

/* ------------------------------------------------------------------ output
 * The screenshot of main.rs:1543-1556 as an 8-bit RGB PNG (rgb = height * width * 3 bytes, row 0 = top; take them from
 * rt_resolve_rgb8), and resumable accumulation: the sums of rt_render with the number of samples they hold.  A load checks
 * that the file belongs to the same frame (size, spp, seed) and its checksum. */
int rth_write_png_rgb8(const char *path, int32_t width, int32_t height, const uint8_t *rgb);
int rth_checkpoint_save(const char *path, int32_t width, int32_t height, int32_t samples_per_pixel, int32_t spp_done, uint32_t seed,
                        const float *rgba_sum);
int rth_checkpoint_load(const char *path, int32_t width, int32_t height, int32_t samples_per_pixel, uint32_t seed, float *rgba_sum,
                        int32_t *spp_done);

/* ------------------------------------------------------------------ input
 * ImageTexture::new (image_texture.rs:20-38): `image::io::Reader::open(p).decode().into_rgba8()` — a texture file to
 * width * height * 4 bytes RGBA, row 0 = top, for rth_add_image_texture / rth_scene_assets.  Baseline JPEG (grey or
 * YCbCr, any chroma sampling, restart intervals) and non-interlaced 8/16-bit PNG; anything else -> RT_E_UNSUPPORTED_PRIMITIVE.
 * The JPEG arithmetic restates jpeg-decoder 0.3.0 (Cargo.lock; not under /root/reference): parity unpinned, see
 * csrc/rt_host_image.cpp.  *rgba is malloc'ed by the library: release it with rth_image_free. */
int  rth_image_decode(const uint8_t *data, size_t size, int32_t *width, int32_t *height, uint8_t **rgba);
int  rth_image_decode_file(const char *path, int32_t *width, int32_t *height, uint8_t **rgba);
void rth_image_free(uint8_t *rgba);

/* WorkBlock list of main.rs:1265-1282 (no shuffle): writes up to cap entries of x0,x1,y0,y1
 * and returns the number of blocks. */
int32_t rth_work_blocks(int32_t width, int32_t height, int32_t block_pixels, int32_t *out_xyxy, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif
