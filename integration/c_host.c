/*
 * c_host.c — a plain C host of the C-ABI (include/rt_api.h, include/rt_host.h): what the reference's Rust side would do through
 * its `extern "C"` block (integration/gpu_ffi.rs), written in the one language this image can compile besides C++.
 *
 *   c_host CONFIG.ron SCENE WIDTH SPP OUT.f32 [noalloc]
 *
 * loads the RON config (main.rs:1241-1263), builds SCENE with the host library (main.rs:106-1133), flattens it, uploads it,
 * renders the frame in two sample ranges (the second call continues the first), resolves it (main.rs:1363-1388) and writes
 * W*H*4 f32 sums followed by W*H*4 f32 resolved pixels to OUT.f32.  It also asks the parity hook for the primary ray through
 * the image centre.  Exit code 0 = every call returned RT_OK.  tests/test_gpu_c_host.py compares the file with the ctypes path.
 */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_host.h"

/* the layout the Rust #[repr(C)] mirrors in integration/gpu_ffi.rs are written against */
_Static_assert(sizeof(rt_camera) == 96, "rt_camera");
_Static_assert(offsetof(rt_render_params, background) == 28 && offsetof(rt_render_params, tile_x0) == 44, "rt_render_params head");
_Static_assert(offsetof(rt_render_params, camera) == 60 && offsetof(rt_render_params, out_location) == 156, "rt_render_params.camera");
_Static_assert(offsetof(rt_render_params, stream) == 160 && offsetof(rt_render_params, cancel) == 168, "rt_render_params.stream");
_Static_assert(offsetof(rt_render_params, pixel_rays) == 184 && offsetof(rt_render_params, projection) == 192, "rt_render_params tail");
_Static_assert(sizeof(rt_render_params) == 208, "rt_render_params");
_Static_assert(sizeof(rt_sampler_desc) == 32, "rt_sampler_desc");
_Static_assert(offsetof(rt_scene_desc, slots) == 32 && offsetof(rt_scene_desc, texels) == 160 && offsetof(rt_scene_desc, matrices) == 208, "rt_scene_desc");
_Static_assert(offsetof(rt_scene_desc, far_limit) == 240 && offsetof(rt_scene_desc, bvh_leaf_max) == 236, "rt_scene_desc tail");
_Static_assert(sizeof(rt_scene_desc) == 248 && sizeof(rt_scene_info) == 28 && sizeof(rt_stats) == 288 && sizeof(rth_user_config) == 100, "sizes");

#define CHECK(call)                                                                                     \
    do {                                                                                                \
        int st_ = (call);                                                                               \
        if (st_ != RT_OK) {                                                                             \
            fprintf(stderr, "%s -> %d: %s | %s\n", #call, st_, ctx ? rt_last_error(ctx) : rt_last_error(NULL), rth_last_error()); \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)

int main(int argc, char **argv) {
    rt_ctx *ctx = NULL;
    if (argc < 6) { fprintf(stderr, "usage: %s CONFIG.ron SCENE WIDTH SPP OUT.f32 [noalloc]\n", argv[0]); return 2; }
    const char *scene_name = argv[2];
    const int width = atoi(argv[3]), spp = atoi(argv[4]);
    const int use_noalloc = argc > 6 && !strcmp(argv[6], "noalloc");

    rth_config *cfg = NULL;
    CHECK(rth_config_load(argv[1], &cfg));
    rth_user_config uc;
    CHECK(rth_config_scene_params(cfg, scene_name, &uc));
    uc.image_width = width;
    uc.samples_per_pixel = spp;
    const int height = rth_image_height(&uc);

    rth_scene *scene = rth_scene_new();
    CHECK(rth_build_scene(scene, scene_name, 0x5EED, NULL));
    rth_flat *flat = NULL;
    CHECK(rth_flatten(rth_scene_graph(scene), &flat));
    rt_scene_desc sd;
    rth_flat_desc(flat, &sd);
    rth_sampler *sampler = NULL;
    CHECK(rth_sampler_new(spp, 0, 0x5EED, &sampler));
    rt_sampler_desc smp;
    rth_sampler_desc(sampler, &smp);

    if (rt_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 3; }
    CHECK(rt_create(0, &ctx));
    CHECK(rt_scene_upload(ctx, &sd));
    CHECK(rt_sampler_upload(ctx, &smp));
    rt_scene_info info;
    info.struct_size = sizeof info;
    CHECK(rt_scene_query(ctx, &info));

    rt_render_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.image_width = width; p.image_height = height;
    p.samples_per_pixel = spp;
    p.max_ray_depth = uc.max_ray_depth;
    memcpy(p.background, uc.background, sizeof p.background);
    p.seed = 0x5EED;
    p.tile_x1 = width; p.tile_y1 = height;
    rth_camera_new(uc.look_from, uc.look_at, uc.world_up, uc.vertical_fov, uc.aspect_ratio, uc.aperture, uc.focus_dist, 0.0f, 1.0f, &p.camera);
    p.out_location = RT_MEM_HOST;
    p.projection = uc.projection;
    p.psi_max = uc.psi_max * 3.14159265358979323846f / 180.0f;       /* RaytracerParams::from, main.rs:1199-1200 */
    p.lambda_max = uc.lambda_max * 3.14159265358979323846f / 180.0f;

    const size_t n_px = (size_t)width * (size_t)height;
    float *sum = calloc(n_px * 4, sizeof(float)), *rgba = calloc(n_px * 4, sizeof(float));
    if (!sum || !rgba) return 4;
    p.spp_begin = 0; p.spp_count = spp / 2;                           /* progressive: two calls add up to the frame */
    CHECK(rt_render(ctx, &p, sum, NULL));
    p.spp_begin = spp / 2; p.spp_count = spp - spp / 2;
    rt_stats stats;
    p.collect_stats = 1;
    CHECK(rt_render(ctx, &p, sum, &stats));
    CHECK(rt_resolve(ctx, sum, (int32_t)n_px, spp, rgba, RT_MEM_HOST, NULL));

    /* parity hook: the ray through the image centre (no lens offset) */
    float ray[7];
    for (int a = 0; a < 3; ++a) {
        ray[a] = p.camera.origin[a];
        ray[3 + a] = p.camera.lower_left_corner[a] + 0.5f * p.camera.horizontal[a] + 0.5f * p.camera.vertical[a] - p.camera.origin[a];
    }
    ray[6] = 0.0f;
    uint32_t prim = 0; float t = 0.0f, n3[3] = {0, 0, 0}, uv[2] = {0, 0}; uint8_t ff = 0;
    if (use_noalloc) CHECK(rt_trace_first_hit_noalloc(ctx, ray, 1, 0.001f, INFINITY, &prim, &t, n3, &ff, uv));
    else CHECK(rt_trace_first_hit(ctx, ray, 1, 0.001f, INFINITY, &prim, &t, n3, &ff, uv));

    FILE *f = fopen(argv[5], "wb");
    if (!f) return 5;
    fwrite(sum, sizeof(float), n_px * 4, f);
    fwrite(rgba, sizeof(float), n_px * 4, f);
    fclose(f);
    printf("{\"width\": %d, \"height\": %d, \"spp\": %d, \"slots\": %u, \"bvh_nodes\": %u, \"rays_second_half\": %llu, \"centre_prim\": %u, \"centre_t\": %.9g, \"front_face\": %d}\n",
           width, height, spp, info.n_slots, info.n_bvh_nodes, (unsigned long long)stats.rays, prim, t, (int)ff);

    free(sum); free(rgba);
    rt_destroy(ctx);
    rth_sampler_free(sampler);
    rth_flat_free(flat);
    rth_scene_free(scene);
    rth_config_free(cfg);
    return 0;
}
