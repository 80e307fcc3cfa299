/*
 * rt_api.h — C-ABI of the MI355X path-tracing backend (librtiow_hip.so).
 *
 * This is the drop-in boundary.  The reference has no plugin API; the seam this
 * library replaces is the worker pool of RaytracerState::new
 * (raytracer/src/main.rs:1326-1409): per pixel `Camera::raytrace_pixel`
 * (raytracer/src/camera.rs:186-251), the gamma/validity pass (main.rs:1363-1388)
 * and the store into `image_pixels[y*W+x]` (main.rs:1461-1466).  Inputs are the same
 * things that loop reads: RaytracerParams (main.rs:1157-1176), the camera
 * (camera.rs:25-36), `world`, `lights` and the sampler table (sampling.rs:11-24) —
 * here flattened by the host into the plain arrays below.
 *
 * Conventions: every function returns rt_status (0 = OK, negative = error) and
 * never throws or aborts; rt_last_error() gives the text.  All pointers in the
 * *_desc structs are HOST pointers owned by the caller for the duration of the
 * call only — the library copies what it keeps.  A context drives one GPU from one
 * host thread at a time.  Image buffers are row-major, row 0 = top, 4 x f32 per
 * pixel (math/src/colors.rs:53-63).
 */
#ifndef RT_API_H
#define RT_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

typedef enum rt_status {
    RT_OK                      =  0,
    RT_E_INVALID               = -1,  /* bad argument / inconsistent descriptor        */
    RT_E_HIP                   = -2,  /* a HIP runtime call failed                      */
    RT_E_RCCL                  = -3,  /* RCCL missing, or one of its calls failed (rt_comm_*, rt_group_*) */
    RT_E_OOM                   = -4,  /* device allocation failed                       */
    RT_E_UNSUPPORTED_PRIMITIVE = -5,  /* scene uses something the kernels do not cover  */
    RT_E_NO_DEVICE             = -6,  /* no usable GPU                                  */
    RT_E_CANCELLED             = -7   /* *cancel became non-zero between batches        */
} rt_status;

/* ---------------------------------------------------------------- flattened scene */

/* Primitive slot type (word 4, bits 0-3 of a slot). */
enum rt_prim_type {
    RT_PRIM_SPHERE        = 0,
    RT_PRIM_MOVING_SPHERE = 1,
    RT_PRIM_XY_RECT       = 2,
    RT_PRIM_XZ_RECT       = 3,
    RT_PRIM_YZ_RECT       = 4,
    RT_PRIM_TRIANGLE      = 5,
    RT_PRIM_CYLINDER      = 6,   /* the quadrics are tested for every ray (always-list), never through the BVH: */
    RT_PRIM_CONE          = 7,   /* cone / paraboloid / hyperboloid ignore t_min (cone.rs:61-65), so a box test  */
    RT_PRIM_PARABOLOID    = 8,   /* over [t_min, t_max] could cull a hit the reference accepts                  */
    RT_PRIM_HYPERBOLOID   = 9
};
#define RT_N_PRIM_TYPES 10

#define RT_SLOT_FLOATS   16   /* one primitive slot = 64 B                        */
#define RT_NODE_FLOATS   16   /* one 4-wide BVH node = 64 B                       */
#define RT_ATTR_FLOATS   16   /* triangle shading attributes = 64 B               */
#define RT_REC_FLOATS     8   /* material / texture / xform / medium / light rec  */
#define RT_MATRIX_FLOATS 36   /* one general Transform: three 3x4 matrices        */
#define RT_PERLIN_WORDS 1536  /* 256*3 f32 gradient + 3*256 i32 permutations      */
#define RT_BVH_EMPTY  0x7fffffff
#define RT_MISS       0xffffffffu

/*
 * Layouts (all little-endian 32-bit words; "i:" = int32 bit pattern in a float lane)
 *
 * BVH node  4 children, their boxes quantised to 8 bits on the grid of the node's own box:
 *           [0..2] origin  [3..5] scale  (plane = origin + scale * q, q = 0..255; child boxes are rounded
 *           OUTWARDS, so the quantised box contains the child's)
 *           [6] u32: min.x q of children 0..3 (child c in byte c)  [7] max.x  [8] min.y  [9] max.y
 *           [10] min.z  [11] max.z  [12..15] i: child refs
 *           ref >= 0: inner node index;  ref < 0: leaf, ~ref = first_slot<<3 | (count-1);
 *           ref == RT_BVH_EMPTY: no child (its box is ignored)
 *
 * slot      [0..3]  q0 geometry   [4] i: type | flip<<4 | (xform+1)<<8
 *           (flip = FlipFace parity that survives Translate's HitRecord::new, transform.rs:36-44)
 *           [5] i: material  [6] i: primitive id (rt_scene_graph.h numbering)
 *           [7] type scalar  [8..11] q2 geometry  [12..15] 0
 *   sphere         q0 = cx cy cz r
 *   moving sphere  q0 = c0x c0y c0z r   [7] = time0   q2 = c1x c1y c1z time1
 *   rects          q0 = a0 a1 b0 b1     [7] = k         (XY: a=x b=y, XZ: a=x b=z, YZ: a=y b=z)
 *   triangle       q0 = p1x p1y p1z p2x [7] = p3z  q2 = p2y p2z p3x p3y
 *   cylinder       q0 = radius zmin zmax phi_max           (cylinder.rs:12-19)
 *   cone           q0 = radius height 0 phi_max            (cone.rs:12-17)
 *   paraboloid     q0 = radius zmin zmax phi_max           (paraboloid.rs:10-16)
 *   hyperboloid    q0 = p1x p1y p1z phi_max  [7] = zmin  q2 = p2x p2y p2z zmax  [12] ah [13] ch
 *                  (p1, p2 after the constructor's swap, hyperboloid.rs:34)
 * tri attr  n1.xyz n2.xyz n3.xyz uv1.xy uv2.xy uv3.xy 0        (indexed by slot)
 * xform     [0] i: bit0 rotate-y, bit1 translate  [1] sin [2] cos [3..5] offset
 *           meaning Translate(RotateY(obj)) (transform.rs:9-153);
 *           bit2 = general Transform (transform.rs:155-213), then [1] i: index into `matrices`;
 *           [0] = 8: a chain, [1] i: first [2] i: count (2..64) = that many consecutive records of the kinds above,
 *           outermost wrapper first: any other nesting of the three wrappers (each hit() hands its ray to its child)
 * matrix    [0..11] world2object rows 0..2 (a00 a01 a02 a03 | a10 ..)  [12..23] obj2world  [24..35] normal2world
 *           (= adjoint(world2object) transposed, transform.rs:171); last rows are 0,0,0,1 and not stored
 * medium    [0..3] boundary sphere c,r  [4] neg_inv_density  [5] i: material  [6] i: prim id  [7] i: 0
 *           or, for any other boundary (a Block / rectangle / sphere under any nesting of Translate / RotateY / Transform) and for a
 *           medium that is itself under such wrappers: [7] i: 1 and
 *           [0] i: first boundary slot  [1] i: number of boundary slots — slots behind the always-list that no tree or
 *           list refers to (primitive id -1); boundary.hit() is the list fold over them (constant_medium.rs:52-58).
 *           [2] i: 1 + the xform record of the medium's OWN wrappers (0: none); the boundary slots' records then begin with
 *           the same wrappers, and the medium measures the ray and makes its hit point in their frame (constant_medium.rs:76-93)
 * light     [0] i: rt_prim_type (sphere, moving sphere or rect)  [1..5] q0 (+k for rects);  type 255 = an object that
 *           keeps the trait defaults of hittable.rs:47-53 (pdf_value 0, random = unit_x).  Moving sphere (objects/sphere.rs:204-226):
 *           [1..3] center(0), where pdf_value tests the ray  [4] radius  [5..7] center(time1) - center(time0), what random aims at
 *           Cylinder (cylinder.rs:134-161): [1] zmin [2] zmax [3] phi_max [4] radius [5..7] centre of its box, what random aims at.
 *           A HittableList inside `lights` (hittable_list.rs:59-75): type 254, [1] i: first [2] i: count (>= 1) = its members, a
 *           run of records behind this one; `lights` with such entries starts with a record of type 253, [1] i: first
 *           [2] i: count = the members of `lights` itself (otherwise they are records 0 .. n_lights-1); at most 4 lists deep
 * material  [0] i: rtg_material_type [1] i: texture [2..4] colour [5] fuzz [6] ior
 * texture   [0] i: rtg_texture_type [1..3] colour [4] i: odd | perlin | texel offset
 *           [5] i: even | width  [6] repeat | scale  [7] i: height
 */
typedef struct rt_scene_desc {
    uint32_t struct_size;           /* sizeof(rt_scene_desc), for versioning */
    const float   *bvh_nodes;       uint32_t n_bvh_nodes;
    int32_t        bvh_root_ref;    /* RT_BVH_EMPTY when nothing is in the BVH */
    uint32_t       bvh_stack_depth; /* entries a traversal stack needs (host computes it) */
    const float   *slots;           uint32_t n_slots;
    uint32_t       always_begin;    /* slots [always_begin, always_begin+always_count) are  */
    uint32_t       always_count;    /* tested for every ray (top-level objects with no usable box) */
    const float   *tri_attrs;       /* n_slots * RT_ATTR_FLOATS, or NULL if no triangles */
    const float   *xforms;          uint32_t n_xforms;
    const float   *media;           uint32_t n_media;    /* in world-list order */
    const float   *lights;          uint32_t n_lights;   /* in lights-list order */
    const float   *materials;       uint32_t n_materials;
    const float   *textures;        uint32_t n_textures;
    const uint32_t *perlin;         uint32_t n_perlin;   /* tables of RT_PERLIN_WORDS */
    const uint8_t *texels;          uint64_t n_texel_bytes; /* RGBA8 atlas */
    const uint32_t *moving_slots;   uint32_t n_moving_slots; /* slot indices of every MovingSphere: tested
                                       directly when ray.time is outside the shutter interval the BVH boxes cover
                                       (isotropic.rs:32 puts hit.t into ray.time) */
    const uint32_t *rect_slots;     uint32_t n_rect_slots;   /* slot indices of every rectangle: the only leaves whose
                                       hit test can ACCEPT a NaN distance for a finite ray (0/0 when the ray lies in the
                                       plane, rectangles.rs:23-34); scanned when a direction component is exactly 0 */
    uint32_t       n_prims;         /* number of primitive ids in the scene */
    const float   *matrices;        uint32_t n_matrices;     /* RT_MATRIX_FLOATS each */
    /* Build the BVH on the GPU instead of taking bvh_nodes (rt_bvh_build.hip): slots
       [0, always_begin) are given in any order with one conservative box each (lo xyz, hi xyz) and are re-ordered
       by the build; the always-list must be the tail of the slot array.  bvh_nodes / bvh_root_ref /
       bvh_stack_depth are ignored.  bvh_leaf_max: 1 or 2 slots per leaf, 0 = the library's choice. */
    const float   *slot_boxes;      uint32_t build_bvh_on_device;   /* rt_bvh_build */   uint32_t bvh_leaf_max;
    /* Rays that start further than this (on some axis) from the middle of the scene's spheres see what HittableList::hit sees
       (hittable_list.rs:39-50): in f32, sphere.rs:46-66 accepts small spheres thousands of radii away by cancellation and the
       reference's flat `world` takes those hits, which a walk that culls by the spheres' boxes would never test.  For such a
       ray the kernels grow every box by the distance at which that can happen (rt_kernels.h, ray_fat), so no leaf whose exact
       test could accept the ray is culled.  The host flatten sets 2^10 x the smallest sphere radius when `world` holds no
       BvhNode and +inf when it does (the reference then culls by boxes itself, bvh.rs:73-91); 0 = the library derives it
       from the slots as for a flat list. */
    float          far_limit;
} rt_scene_desc;

/* rt_scene_desc.build_bvh_on_device: who builds the tree.  RT_BVH_DEVICE = parallel locally-ordered clustering over Morton-sorted
 * leaves (close to the host's SAH tree); RT_BVH_DEVICE_LBVH = Morton radix tree (fastest build, a third more node visits). */
enum rt_bvh_build { RT_BVH_HOST = 0, RT_BVH_DEVICE = 1, RT_BVH_DEVICE_LBVH = 2 };

/* What rt_scene_upload left on the device (the BVH may have been built there). */
typedef struct rt_scene_info {
    uint32_t struct_size;
    uint32_t n_slots, n_bvh_nodes, bvh_stack_depth;
    int32_t  bvh_root_ref;
    uint32_t built_on_device;
    float    bvh_build_ms;          /* GPU time of the build (HIP events), 0 for a host-built tree */
} rt_scene_info;

/* Sampler table: SamplerBase<MultiJittered> (raytracer/src/sampling.rs:11-24, 137-163). */
typedef struct rt_sampler_desc {
    uint32_t     sets;              /* 83 in the reference (sampling.rs:31) */
    uint32_t     samples_in_set;    /* = samples_per_pixel */
    const float *samples;           /* sets*samples_in_set * 2 : unit square */
    const float *disk_samples;      /* sets*samples_in_set * 2 : unit disk   */
    const uint32_t *shuffled_indices; /* sets*samples_in_set */
} rt_sampler_desc;

/* Camera block = the fields of `Camera` (raytracer/src/camera.rs:25-36). */
typedef struct rt_camera {
    float origin[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float u[3], v[3], w[3];
    float lens_radius;
    float time0, time1;
} rt_camera;

enum rt_mem_location { RT_MEM_HOST = 0, RT_MEM_DEVICE = 1 };

/* Projection (main.rs:1140-1146); the numbering of the RON config's enum */
enum rt_projection { RT_PROJ_PERSPECTIVE = 0, RT_PROJ_ORTHOGRAPHIC = 1, RT_PROJ_FISHEYE = 2, RT_PROJ_SPHERICAL_PANORAMIC = 3 };

#define RT_MAX_RAY_DEPTH 65535
typedef struct rt_render_params {
    uint32_t  struct_size;
    int32_t   image_width, image_height;     /* full frame (main.rs:1160-1161)                 */
    int32_t   samples_per_pixel;             /* spp of the whole frame = sampler samples_in_set */
    int32_t   spp_begin, spp_count;          /* sample range rendered by THIS call             */
    int32_t   max_ray_depth;                 /* main.rs:1164; <= RT_MAX_RAY_DEPTH (the reference's i32 is unbounded: beyond
                                                that rt_render returns RT_E_INVALID); <= 0 renders black, camera.rs:258-262 */
    float     background[3];                 /* main.rs:1172                                   */
    uint32_t  seed;                          /* counter-RNG key; the reference is unseeded     */
    int32_t   tile_x0, tile_y0, tile_x1, tile_y1; /* pixel rectangle, x1/y1 exclusive; 0,0,W,H = all */
    rt_camera camera;
    int32_t   out_location;                  /* rt_mem_location of out_rgba_sum                */
    void     *stream;                        /* hipStream_t, NULL = default stream             */
    const volatile int32_t *cancel;          /* polled between batches (main.rs:1338), may be NULL */
    int32_t   collect_stats;                 /* 1 = run the counting variant and fill rt_stats */
    uint32_t *pixel_rays;                    /* optional W*H counters (same location as the image): += rays
                                                traced for that pixel; only with collect_stats = 1 */
    int32_t   projection;                    /* rt_projection (camera.rs:198-249)              */
    float     psi_max, lambda_max;           /* radians (main.rs:1199-1200); FishEye / SphericalPanoramic */
} rt_render_params;

/* Work counters (SURVEY.md §8(d) FLOP model inputs).  Only filled when collect_stats=1. */
typedef struct rt_stats {
    uint64_t paths;            /* pixel samples started                                 */
    uint64_t rays;             /* closest-hit queries issued by ray_color (camera.rs:264) */
    uint64_t node_visits;      /* BVH nodes fetched                                     */
    uint64_t box_tests;        /* child boxes tested: the non-empty children (<= 4) of every visited node */
    uint64_t prim_tests[RT_N_PRIM_TYPES]; /* by rt_prim_type, traversal + always-list   */
    uint64_t medium_tests;     /* ConstantMedium::hit evaluations                       */
    uint64_t light_pdf_evals;  /* lights.pdf_value object evaluations                   */
    uint64_t shades[5];        /* scatter calls by rtg_material_type                    */
    uint64_t tex_evals[4];     /* texture value() calls by rtg_texture_type             */
    uint64_t accepted_hits;    /* rays that found a hit                                 */
    /* Lane occupancy of the render kernel's phases (0 node step, 1 leaf step, 2 shade, 3 path start,
       4 begin-trace): phase_slots += 64 each time a wave runs the phase, phase_lanes += lanes working in it. */
    uint64_t phase_lanes[5];
    uint64_t phase_slots[5];
} rt_stats;

/* ------------------------------------------------------------------- entry points */

/* Number of HIP devices visible to this process (0 if none). */
int rt_device_count(void);

/* Create a context on one device.  Replaces the thread-pool set-up, main.rs:1314-1336. */
int rt_create(int device_id, rt_ctx **out);
void rt_destroy(rt_ctx *ctx);

/* Text of the last error on this context (ctx may be NULL: last error of rt_create). */
const char *rt_last_error(const rt_ctx *ctx);

/* Copy the flattened `world` + `lights` to the device.
 * Replaces `Arc::new(world)`, `Arc::new(lights)` handed to the workers, main.rs:1320-1333. */
int rt_scene_upload(rt_ctx *ctx, const rt_scene_desc *scene);

/* Facts about the uploaded scene, and (diagnostics / tests) a copy of its BVH nodes and slots as the kernels read them:
 * nodes = n_bvh_nodes * RT_NODE_FLOATS floats, slots = n_slots * RT_SLOT_FLOATS floats (either may be NULL). */
int rt_scene_query(rt_ctx *ctx, rt_scene_info *info);
int rt_scene_read_bvh(rt_ctx *ctx, float *nodes, float *slots);

/* Copy the sampler table to the device.  Replaces `s.clone()` per worker, main.rs:1324,1335. */
int rt_sampler_upload(rt_ctx *ctx, const rt_sampler_desc *sampler);

/* Path-trace samples [spp_begin, spp_begin+spp_count) of every pixel of the tile and ADD the
 * radiance sums into out_rgba_sum (W*H*4 f32, un-normalised; alpha += spp_count).
 * Replaces the worker loop + Camera::raytrace_pixel, main.rs:1337-1361, camera.rs:186-309.
 * With out_location = RT_MEM_DEVICE the work is ordered on params->stream (it starts after what the
 * caller queued there, and what the caller queues next sees the image); the call itself returns once
 * the frame's last kernels are queued — the host feeds the pipeline, so for the host it lasts nearly as
 * long as the render, less the frame's last launches (the few paths still alive then: a fifth of a
 * 10-spp frame of Book-1).  `cancel` is polled between groups of eight iterations.
 * FRAMES IN FLIGHT: the context keeps two workspaces in device memory (each: path pool 1.2 GB — 4.8 GB
 * once a call of 2^28 items or more was made on a scene with a tree; a pool the device cannot give is
 * tried again at a quarter of the size —, sample buffer up to RT_WF_SAMPLES_MB = 8192 MB; grown on demand,
 * kept until rt_destroy) and uses them in turn,
 * so a call ordered on ANOTHER stream than the one before starts while that one's last launches still
 * run (rt_stream_create; different output buffers, or the caller orders the calls: the sums are added
 * into out_rgba_sum by the frame's last kernel).  Calls on one stream run one after the other as before.
 * A third call first waits for the first to end; rt_scene_upload, rt_sampler_upload and rt_destroy wait
 * for every frame in flight.  The one thing the host no longer sees before it returns is the pipeline's last
 * status (an empty pool, every item handed out — what the bound on its iterations guarantees): the call
 * that next waits for that frame (the render after next, an upload, rt_synchronize) checks it and
 * reports RT_E_HIP if it does not hold.  (HIP gives a process 4 hardware queues by default and then lets the streams
 * of two frames share them, which serialises the frames: set GPU_MAX_HW_QUEUES=8 in the environment.) */
int rt_render(rt_ctx *ctx, const rt_render_params *params, float *out_rgba_sum, rt_stats *stats);

/* The counters of the last rt_render with collect_stats = 1, split by the kernel that did the work: 0 generate (camera rays and
 * the part of world.hit() that precedes the walk, for new paths), 1 traverse (the BVH walk), 2 shade (ray_color's body and the
 * next ray's start), 3 tail (all three for the last paths of a batch).  The four add up to rt_render's rt_stats.  Measurement only —
 * NO COUNTERPART IN THE REFERENCE (it has no counters): bench.py prices each kernel's work against that kernel's time. */
int rt_last_kernel_stats(rt_ctx *ctx, int32_t kernel, rt_stats *out);

/* Gamma / validity pass, main.rs:1363-1388:  c = clamp(sqrt(sum/spp),0,1), !is_normal(sum) -> 0,
 * alpha copied.  location applies to both pointers; stream as in rt_render (NULL = default). */
int rt_resolve(rt_ctx *ctx, const float *rgba_sum, int32_t n_pixels, int32_t spp,
               float *out_rgba, int32_t location, void *stream);

/* The screenshot's pixel format (main.rs:1543-1551: `DynamicImage::ImageRgb32F(..).to_rgb8()`): the gamma pass above, then
 * image 0.24.6's f32 -> u8 conversion, round(clamp(c, 0, 1) * 255) with halves away from zero.  out_rgb = n_pixels * 3 bytes.
 * (The reference hands to_rgb8 the first W*H*3 floats of its RGBA buffer, so its own file is scrambled — SURVEY N6; this
 * writes the image that call means.) */
int rt_resolve_rgb8(rt_ctx *ctx, const float *rgba_sum, int32_t n_pixels, int32_t spp, uint8_t *out_rgb, int32_t location, void *stream);

/* Parity hook: closest hit of `world` for n given rays (HittableList::hit semantics,
 * hittable_list.rs:39-50, without the stochastic ConstantMedium objects).
 * rays_soa = 7 planes of n floats: ox oy oz dx dy dz time (math/src/ray.rs:10-14).
 * prim_id = RT_MISS on miss.  normal3 = n x 3 (face-forwarded HitRecord.normal), uv2 = n x 2
 * (may be NULL), front_face = n bytes.  All pointers are host pointers. */
int rt_trace_first_hit(rt_ctx *ctx, const float *rays_soa, int32_t n, float t_min, float t_max,
                       uint32_t *prim_id, float *t, float *normal3, uint8_t *front_face, float *uv2);

/* rt_trace_first_hit for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc,
 * hosts that must not touch the allocator while something else is being captured into a graph).  NO COUNTERPART IN THE
 * REFERENCE.  Same arguments, same kernel, same results; the staging of rays and results and the traversal-stack overflow live
 * in a static device array of RT_NOALLOC_SCRATCH_BYTES per GPU (held from library load to process exit) instead of a buffer
 * allocated per call, and the call makes no hipMalloc / hipFree.  It is SYNCHRONOUS like rt_trace_first_hit — host pointers in,
 * copies on the library's own stream, a stream synchronize before it returns — so it is not itself capturable into a graph.
 * n is limited by that array: 57 bytes per ray plus 4 bytes per ray and BVH stack level beyond 48 (about 147 000 rays for the
 * BASELINE scenes); a larger batch returns RT_E_INVALID with the limit in rt_last_error() and touches nothing.
 * All contexts of one GPU share its array: the library serialises the calls of one process on one GPU (a mutex per device; the
 * call blocks anyway); calls from different PROCESSES have their own copies of the array. */
#define RT_NOALLOC_SCRATCH_BYTES (8u << 20)
int rt_trace_first_hit_noalloc(rt_ctx *ctx, const float *rays_soa, int32_t n, float t_min, float t_max,
                               uint32_t *prim_id, float *t, float *normal3, uint8_t *front_face, float *uv2);

/* ------------------------------------------------------------------- multi-GPU
 * No counterpart in the reference (one process, one thread pool: main.rs:1314-1409).  The frame is sharded over GPUs by
 * sample range (rt_render_params.spp_begin / spp_count; the RNG is keyed by the absolute sample index, so the image does not
 * depend on how many GPUs rendered it) or by tile rectangle, every GPU accumulates into its own W*H*4 f32 buffer, and ONE
 * RCCL reduce (sum, in place, to the root; xGMI gives every GPU a direct link to it) makes the frame.  RCCL is loaded on first
 * use; without it these calls return RT_E_RCCL and everything else keeps working.
 *
 * One process per GPU (MPI / torchrun style): rank 0 calls rt_comm_unique_id and hands the 128 bytes to the other ranks by
 * whatever channel the host has; every rank calls rt_comm_create (collective: returns when all `world` ranks have called it),
 * renders its share with rt_render into a DEVICE buffer, then calls rt_comm_reduce_to_root on the same stream.  The call is
 * asynchronous like the render; on `root` the buffer then holds the sum of all ranks' buffers, elsewhere it is unchanged. */
#define RT_COMM_ID_BYTES 128
#include <stddef.h>
/* Device buffers for a host that has no HIP binding of its own (the Rust host): an accumulation image on the context's GPU,
 * zeroed / read back in stream order (stream: hipStream_t, NULL = default; rt_device_read waits for the copy). */
int rt_device_alloc(rt_ctx *ctx, size_t bytes, void **out_device_ptr);
int rt_device_free(rt_ctx *ctx, void *device_ptr);
int rt_device_zero(rt_ctx *ctx, void *device_ptr, size_t bytes, void *stream);
int rt_device_read(rt_ctx *ctx, void *dst_host, const void *src_device, size_t bytes, void *stream);
/* ... and streams for it (hipStream_t as void*, non-blocking): what rt_render, rt_resolve, rt_device_* and the reduces take as
 * `stream`.  rt_stream_wait orders what is queued on `stream` from now on behind what is queued on `after` so far (NULL = the
 * default stream).  rt_stream_destroy waits for the stream's work.  NO COUNTERPART IN THE REFERENCE (its frames are one
 * thread-pool job each): two streams are what lets two frames be in flight (rt_render). */
int rt_stream_create(rt_ctx *ctx, void **out_stream);
int rt_stream_destroy(rt_ctx *ctx, void *stream);
int rt_stream_wait(rt_ctx *ctx, void *stream, void *after);
typedef struct rt_comm rt_comm;
int rt_comm_unique_id(char id[RT_COMM_ID_BYTES]);
int rt_comm_create(rt_ctx *ctx, const char id[RT_COMM_ID_BYTES], int rank, int world, rt_comm **out);
int rt_comm_reduce_to_root(rt_comm *comm, float *rgba_sum_device, int64_t n_floats, int root, void *stream);
/* (rt_comm_reduce_to_root needs the communicator's context alive — it reports errors there; rt_comm_destroy does not touch the
 * context and may run after rt_destroy.) */
void rt_comm_destroy(rt_comm *comm);

/* One process, n GPUs (what a Rust host with one thread per GPU would use): a group owns one context per listed device and a
 * communicator over them.  rt_group_ctx(g, i) is the context of device_ids[i] (owned by the group: do not rt_destroy it);
 * upload the scene and sampler to each and render from one host thread per context.  rt_group_reduce_to_root takes the n
 * device buffers (rgba_sums[i] on device i) and the n streams the renders were ordered on (NULL = default streams) and sums
 * them into rgba_sums[root]; it returns when the reduce is queued on every stream.  A group of one device needs no RCCL and its
 * reduce is a no-op, so a host can use the group calls whatever the number of GPUs (integration/render_gpu.rs). */
typedef struct rt_group rt_group;
int rt_group_create(const int *device_ids, int n, rt_group **out);
int rt_group_size(const rt_group *g);
rt_ctx *rt_group_ctx(rt_group *g, int i);
int rt_group_reduce_to_root(rt_group *g, float *const *rgba_sums, int64_t n_floats, int root, void *const *streams);
void rt_group_destroy(rt_group *g);

/* Block until everything queued on `stream` (NULL = default) by this context is done. */
int rt_synchronize(rt_ctx *ctx, void *stream);

/* Name, CU count and clock of the context's device (for bench reports). */
int rt_device_info(rt_ctx *ctx, char *name, int32_t name_len, int32_t *cu_count, int32_t *clock_mhz);

#ifdef __cplusplus
}
#endif
#endif
