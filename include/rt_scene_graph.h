/*
 * rt_scene_graph.h — POD scene graph: the reference's object model as plain data.
 *
 * The reference keeps its scene as `Arc<dyn Hittable>` / `Arc<dyn Material>` /
 * `Arc<dyn Texture>` trees with no introspection (raytracer/src/hittable.rs:44-54,
 * material.rs:18-35, texture.rs:3-5).  This header is the same model with the
 * pointers replaced by indices, so that it can cross a C boundary:
 *
 *   - the host library (include/rt_host.h) builds it (what a `Hittable::flatten`
 *     visitor on the Rust side would emit) and lowers it to the SoA/BVH buffers of
 *     include/rt_api.h;
 *   - the CPU oracle (oracle/) consumes it and rebuilds a reference-style object
 *     tree with virtual hit()/scatter().
 *
 * Primitive ids.  Every leaf primitive gets an id by a depth-first walk from the
 * world root, children in stored order:  Sphere/MovingSphere/rect = 1 id,
 * Block = 6 ids in the side order of raytracer/src/block.rs:27-79,
 * ConstantMedium = 1 id (its boundary is not numbered), TriangleMesh = one id per
 * triangle in index order, a quadric = 1 id.  Wrappers (FlipFace, Translate, RotateY,
 * Transform, Bvh, List) take no id.  Host flattener and oracle must both follow this rule.
 */
#ifndef RT_SCENE_GRAPH_H
#define RT_SCENE_GRAPH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Hittable kinds (reference file that defines each is cited). */
enum rtg_node_type {
    RTG_SPHERE          = 0,  /* objects/sphere.rs:10-113   f = cx,cy,cz,r                         */
    RTG_MOVING_SPHERE   = 1,  /* objects/sphere.rs:115-227  f = c0xyz, c1xyz, time0, time1, r      */
    RTG_XY_RECT         = 2,  /* rectangles.rs:12-78        f = x0,x1,y0,y1,k                      */
    RTG_XZ_RECT         = 3,  /* rectangles.rs:82-148       f = x0,x1,z0,z1,k                      */
    RTG_YZ_RECT         = 4,  /* rectangles.rs:152-218      f = y0,y1,z0,z1,k                      */
    RTG_BLOCK           = 5,  /* block.rs:17-109            f = p0xyz, p1xyz                       */
    RTG_FLIP_FACE       = 6,  /* flip_face.rs:5-29          1 child                                */
    RTG_TRANSLATE       = 7,  /* transform.rs:9-55          f = offset xyz, 1 child                */
    RTG_ROTATE_Y        = 8,  /* transform.rs:57-153        f = angle in degrees, 1 child; i[0] = 1: f[1], f[2] = sin, cos given */
    RTG_BVH             = 9,  /* bvh.rs:15-89               children = the slice handed to BvhNode::new */
    RTG_LIST            = 10, /* hittable_list.rs:7-76      children in insertion order            */
    RTG_CONSTANT_MEDIUM = 11, /* constant_medium.rs:13-95   f = density, 1 child (boundary); material = Isotropic */
    RTG_TRIANGLE_MESH   = 12, /* triangle_mesh.rs:15-268    i[0] = mesh index                      */
    RTG_TRANSFORM       = 13, /* transform.rs:155-213       f = rows 0..2 of obj2world (row-major 3x4; the last row
                                                            is 0,0,0,1: every Transform of main.rs is affine), 1 child */
    RTG_CYLINDER        = 14, /* cylinder.rs:12-162         f = radius, zmin, zmax, phi_max        */
    RTG_CONE            = 15, /* cone.rs:12-129             f = radius, height, phi_max            */
    RTG_PARABOLOID      = 16, /* paraboloid.rs:10-145       f = radius, zmin, zmax, phi_max        */
    RTG_HYPERBOLOID     = 17  /* hyperboloid.rs:13-170      f = p1 xyz, p2 xyz (the constructor's arguments), phi_max */
};

typedef struct rtg_node {
    int32_t type;          /* rtg_node_type */
    int32_t material;      /* index into materials, -1 for wrappers */
    int32_t child_begin;   /* first entry in rtg_graph.children */
    int32_t child_count;
    float   f[12];
    int32_t i[4];
} rtg_node;

enum rtg_material_type {
    RTG_MAT_LAMBERTIAN    = 0, /* lambertian.rs   tex = albedo texture          */
    RTG_MAT_METAL         = 1, /* metal.rs        color = albedo, fuzz          */
    RTG_MAT_DIELECTRIC    = 2, /* dielectric.rs   ior                           */
    RTG_MAT_DIFFUSE_LIGHT = 3, /* diffuse_light.rs tex = emit texture           */
    RTG_MAT_ISOTROPIC     = 4  /* isotropic.rs    tex = albedo texture          */
};

typedef struct rtg_material {
    int32_t type;
    int32_t tex;
    float   color[3];
    float   fuzz;
    float   ior;
    int32_t _pad;
} rtg_material;

enum rtg_texture_type {
    RTG_TEX_SOLID   = 0, /* solid_color_texture.rs  color                                */
    RTG_TEX_CHECKER = 1, /* checker_texture.rs      odd, even (texture ids), scale = repeat_factor */
    RTG_TEX_NOISE   = 2, /* noise_texture.rs        scale, perlin = table index          */
    RTG_TEX_IMAGE   = 3  /* image_texture.rs        image = image index (-1 = blank → cyan) */
};

typedef struct rtg_texture {
    int32_t type;
    float   color[3];
    int32_t odd, even;
    float   scale;
    int32_t perlin;
    int32_t image;
    int32_t _pad[3];
} rtg_texture;

/* perlin.rs:5-10 — 256 unit vectors and three permutations of 0..255 */
typedef struct rtg_perlin {
    float   ranvec[256][3];
    int32_t perm_x[256];
    int32_t perm_y[256];
    int32_t perm_z[256];
} rtg_perlin;

/* image_texture.rs:3-8 — RGBA8, row 0 = top, bytes_per_scanline = 4*width */
typedef struct rtg_image {
    int32_t        width, height;
    const uint8_t *rgba;
} rtg_image;

/* World-space vertex as left by TriangleMesh::new (triangle_mesh.rs:53-72). */
typedef struct rtg_vertex {
    float pos[3];
    float normal[3];
    float uv[2];
} rtg_vertex;

/* One GeometryNode of a mesh (geometry_import.rs:24-45): a run of the index array. */
typedef struct rtg_mesh_node {
    uint32_t index_begin;  /* into rtg_mesh.indices, multiple of 3 */
    uint32_t index_count;
} rtg_mesh_node;

typedef struct rtg_mesh {
    const rtg_vertex    *vertices;
    uint32_t             n_vertices;
    const uint32_t      *indices;
    uint32_t             n_indices;
    const rtg_mesh_node *nodes;
    uint32_t             n_nodes;
} rtg_mesh;

typedef struct rtg_graph {
    const rtg_node     *nodes;      uint32_t n_nodes;
    const int32_t      *children;   uint32_t n_children;
    const rtg_material *materials;  uint32_t n_materials;
    const rtg_texture  *textures;   uint32_t n_textures;
    const rtg_perlin   *perlins;    uint32_t n_perlins;
    const rtg_image    *images;     uint32_t n_images;
    const rtg_mesh     *meshes;     uint32_t n_meshes;
    int32_t world_root;   /* RTG_LIST — `world` of scene_*() (main.rs:1301-1312) */
    int32_t lights_root;  /* RTG_LIST — `lights` of scene_*()                   */
} rtg_graph;

#ifdef __cplusplus
}
#endif
#endif
