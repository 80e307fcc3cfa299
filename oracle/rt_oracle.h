/*
 * rt_oracle.h — C-ABI of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library.  The product (librtiow_hip.so, librtiow_host.so) never does.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference has no tests, golden images or seedable RNG
 * for the ray tracer and cannot be compiled here (Rust; no cargo/rustc in
 * the image).  The oracle is a statement-by-statement restatement of the reference's algorithm
 * (citations in rt_oracle.cpp), pinned by hand-derived known answers in tests/golden/.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include "rt_api.h"
#include "rt_scene_graph.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

typedef struct orc_options {
    uint32_t bvh_seed;          /* seeds the random axis of BvhNode::new (bvh.rs:17) */
    int32_t  mesh_nearest;      /* 0 = reference: first triangle in index order per node (triangle_mesh.rs:256-258);
                                   1 = nearest triangle (what a BVH returns; the parity target) */
    int32_t  use_libm;          /* 1 = glibc sinf/cosf/acosf/atan2f/logf instead of include/rt_portable_math.h */
    int32_t  exhaustive;        /* 1 = BvhNode tests every child (no box culling): flat-list semantics */
} orc_options;

int  orc_scene_create(const rtg_graph *g, const orc_options *opt, orc_scene **out);
void orc_scene_destroy(orc_scene *s);
const char *orc_last_error(void);

typedef struct orc_render_params {
    int32_t   image_width, image_height;
    int32_t   samples_per_pixel, spp_begin, spp_count, max_ray_depth;
    float     background[3];
    uint32_t  seed;
    int32_t   tile_x0, tile_y0, tile_x1, tile_y1;
    rt_camera camera;
    int32_t   threads;              /* worker threads (main.rs:1326); tiles of block_pixels^2 in a LIFO queue */
    int32_t   block_pixels;
    int32_t   projection;           /* rt_projection */
    float     psi_max, lambda_max;  /* radians */
} orc_render_params;

/* Adds radiance sums into out_rgba_sum (W*H*4, alpha += spp_count) like rt_render.
 * rays_out (may be NULL) = closest-hit queries issued by ray_color; pixel_rays (may be NULL) = per pixel. */
/* KAT hook: the ray of get_ray_<projection>(u, v) (camera.rs:82-184) with unit-disk lens sample (dx, dy);
 * out6 = origin, direction.  Returns 0 for FishEye's None. */
int orc_project(const orc_render_params *p, float u, float v, float dx, float dy, float *out6);
/* The screenshot's pixels (main.rs:1543-1551): gamma pass, then to_rgb8.  out = n_pixels * 3 bytes. */
void orc_resolve_rgb8(const float *sum, int32_t n_pixels, int32_t spp, uint8_t *out);
/* Debugging aid: print one line per bounce of every traced path on stderr (also: env ORC_TRACE). */
void orc_set_trace(int on);

int orc_render(const orc_scene *s, const orc_render_params *p, const rt_sampler_desc *sampler, float *out_rgba_sum,
               uint64_t *rays_out, uint32_t *pixel_rays);

/* world.hit(r, t_min, t_max) without the stochastic ConstantMedium objects; same outputs as rt_trace_first_hit. */
int orc_first_hit(const orc_scene *s, const float *rays_soa, int32_t n, float t_min, float t_max, uint32_t *prim_id,
                  float *t, float *normal3, uint8_t *front_face, float *uv2);

/* main.rs:1363-1388 */
void orc_resolve(const float *rgba_sum, int32_t n_pixels, int32_t spp, float *out_rgba);

/* Unit hooks for known-answer tests (tests/test_oracle_kat.py). */
int   orc_texture_value(const orc_scene *s, int32_t tex, float u, float v, const float p[3], float out_rgb[3]);
int   orc_aabb_hit(const float box_min[3], const float box_max[3], const float ray7[7], float t_min, float t_max);
void  orc_onb(const float n[3], float out_uvw9[9]);
float orc_schlick(float cosine, float ior);
void  orc_refract(const float uv[3], const float n[3], float etai_over_etat, float out[3]);
float orc_math(int32_t fn /*0 sin 1 cos 2 acos 3 atan2 4 ln*/, float a, float b, int32_t use_libm);
/* PerlinNoise::noise (turbulence_depth = 0) or ::turbulence(p, depth) on a caller-supplied table */
float orc_perlin(const rtg_perlin *table, const float p[3], int32_t turbulence_depth);
/* lights.pdf_value(o, v) and the scattering decision of one material for a given hit (KATs) */
float orc_lights_pdf_value(const orc_scene *s, const float o[3], const float v[3]);

#ifdef __cplusplus
}
#endif
#endif
