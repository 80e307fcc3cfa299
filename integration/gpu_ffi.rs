//! gpu_ffi.rs — `raytracer/src/gpu_ffi.rs`: the Rust side of include/rt_api.h, include/rt_host.h and include/rt_scene_graph.h.
//!
//! UNVERIFIED SOURCE: the build image has no rustc / cargo, so this file has never been compiled.  What IS verified is the
//! layout it is written against: integration/c_host.c carries `_Static_assert`s on the same offsets and sizes and drives the
//! same entry points from C on the GPU box (tests/test_gpu_c_host.py), and tests/test_abi.py checks the ctypes mirrors against
//! `sizeof` as gcc sees the headers.  Every struct below is `#[repr(C)]`, field for field in header order; the trailing
//! comments give the byte offsets gcc reports (x86-64).
//!
//! Link: `println!("cargo:rustc-link-lib=dylib=rtiow_hip"); println!("cargo:rustc-link-lib=dylib=rtiow_host");` in build.rs.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_E_INVALID: c_int = -1;
pub const RT_E_HIP: c_int = -2;
pub const RT_E_RCCL: c_int = -3;
pub const RT_E_OOM: c_int = -4;
pub const RT_E_UNSUPPORTED_PRIMITIVE: c_int = -5;
pub const RT_E_NO_DEVICE: c_int = -6;
pub const RT_E_CANCELLED: c_int = -7;

pub const RT_N_PRIM_TYPES: usize = 10;
pub const RT_MEM_HOST: i32 = 0;
pub const RT_MEM_DEVICE: i32 = 1;
pub const RT_BVH_HOST: u32 = 0;
pub const RT_BVH_DEVICE: u32 = 1;
pub const RT_BVH_DEVICE_LBVH: u32 = 2;
pub const RT_MAX_RAY_DEPTH: i32 = 65535;
pub const RT_COMM_ID_BYTES: usize = 128;
pub const RT_NOALLOC_SCRATCH_BYTES: u32 = 8 << 20;
pub const RTH_FLATTEN_DEVICE_BVH: u32 = 1;

pub enum rt_ctx {}
pub enum rt_comm {}
pub enum rt_group {}
pub enum rth_scene {}
pub enum rth_flat {}
pub enum rth_sampler {}
pub enum rth_config {}
pub enum rth_geometry {}
pub enum rtg_graph {} // only ever passed through (rth_scene_graph -> rth_flatten)

/// rt_scene_desc (rt_api.h): sizeof 248
#[repr(C)]
pub struct rt_scene_desc {
    pub struct_size: u32,            // 0
    pub bvh_nodes: *const f32,       // 8
    pub n_bvh_nodes: u32,            // 16
    pub bvh_root_ref: i32,           // 20
    pub bvh_stack_depth: u32,        // 24
    pub slots: *const f32,           // 32
    pub n_slots: u32,                // 40
    pub always_begin: u32,           // 44
    pub always_count: u32,           // 48
    pub tri_attrs: *const f32,       // 56
    pub xforms: *const f32,          // 64
    pub n_xforms: u32,               // 72
    pub media: *const f32,           // 80
    pub n_media: u32,                // 88
    pub lights: *const f32,          // 96
    pub n_lights: u32,               // 104
    pub materials: *const f32,       // 112
    pub n_materials: u32,            // 120
    pub textures: *const f32,        // 128
    pub n_textures: u32,             // 136
    pub perlin: *const u32,          // 144
    pub n_perlin: u32,               // 152
    pub texels: *const u8,           // 160
    pub n_texel_bytes: u64,          // 168
    pub moving_slots: *const u32,    // 176
    pub n_moving_slots: u32,         // 184
    pub rect_slots: *const u32,      // 192
    pub n_rect_slots: u32,           // 200
    pub n_prims: u32,                // 204
    pub matrices: *const f32,        // 208
    pub n_matrices: u32,             // 216
    pub slot_boxes: *const f32,      // 224
    pub build_bvh_on_device: u32,    // 232
    pub bvh_leaf_max: u32,           // 236
    pub far_limit: f32,              // 240
}

/// rt_scene_info: sizeof 28
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rt_scene_info {
    pub struct_size: u32,
    pub n_slots: u32,
    pub n_bvh_nodes: u32,
    pub bvh_stack_depth: u32,
    pub bvh_root_ref: i32,
    pub built_on_device: u32,
    pub bvh_build_ms: f32,
}

/// rt_sampler_desc: sizeof 32 — the three Vecs of SamplerBase (sampling.rs:15-18)
#[repr(C)]
pub struct rt_sampler_desc {
    pub sets: u32,
    pub samples_in_set: u32,
    pub samples: *const f32,          // Vec<Vec2> as pairs of f32
    pub disk_samples: *const f32,
    pub shuffled_indices: *const u32,
}

/// rt_camera: sizeof 96 — the fields of `Camera` (camera.rs:25-36)
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rt_camera {
    pub origin: [f32; 3],
    pub lower_left_corner: [f32; 3],
    pub horizontal: [f32; 3],
    pub vertical: [f32; 3],
    pub u: [f32; 3],
    pub v: [f32; 3],
    pub w: [f32; 3],
    pub lens_radius: f32,
    pub time0: f32,
    pub time1: f32,
}

/// rt_render_params: sizeof 208
#[repr(C)]
pub struct rt_render_params {
    pub struct_size: u32,             // 0
    pub image_width: i32,             // 4
    pub image_height: i32,            // 8
    pub samples_per_pixel: i32,       // 12
    pub spp_begin: i32,               // 16
    pub spp_count: i32,               // 20
    pub max_ray_depth: i32,           // 24
    pub background: [f32; 3],         // 28
    pub seed: u32,                    // 40
    pub tile_x0: i32,                 // 44
    pub tile_y0: i32,
    pub tile_x1: i32,
    pub tile_y1: i32,
    pub camera: rt_camera,            // 60
    pub out_location: i32,            // 156
    pub stream: *mut c_void,          // 160  hipStream_t, null = default stream
    pub cancel: *const i32,           // 168  &AtomicBool-as-i32 of cancel_token (main.rs:1338), may be null
    pub collect_stats: i32,           // 176
    pub pixel_rays: *mut u32,         // 184
    pub projection: i32,              // 192  Projection as i32 (main.rs:1140-1146)
    pub psi_max: f32,                 // 196  radians, as RaytracerParams holds them
    pub lambda_max: f32,              // 200
}

/// rt_stats: sizeof 288
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rt_stats {
    pub paths: u64,
    pub rays: u64,
    pub node_visits: u64,
    pub box_tests: u64,
    pub prim_tests: [u64; RT_N_PRIM_TYPES],   // 32
    pub medium_tests: u64,                    // 112
    pub light_pdf_evals: u64,
    pub shades: [u64; 5],                     // 128
    pub tex_evals: [u64; 4],                  // 168
    pub accepted_hits: u64,                   // 200
    pub phase_lanes: [u64; 5],                // 208
    pub phase_slots: [u64; 5],                // 248
}

/// rth_user_config (rt_host.h): sizeof 100 — RaytracerUserConfig (main.rs:1135-1154); only needed if the host lets the
/// library parse the RON file instead of serde
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rth_user_config {
    pub workers: i32,
    pub worker_block_pixels: i32,
    pub aspect_ratio: f32,
    pub image_width: i32,
    pub samples_per_pixel: i32,
    pub max_ray_depth: i32,
    pub vertical_fov: f32,
    pub look_from: [f32; 3],
    pub look_at: [f32; 3],
    pub world_up: [f32; 3],
    pub aperture: f32,
    pub focus_dist: f32,
    pub shuffle_workblocks: i32,
    pub background: [f32; 3],
    pub projection: i32,
    pub psi_max: f32,
    pub lambda_max: f32,
}

/// rtg_vertex / rtg_mesh_node / rtg_perlin (rt_scene_graph.h) as the mesh and noise constructors take them
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rtg_vertex { pub pos: [f32; 3], pub normal: [f32; 3], pub uv: [f32; 2] }
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rtg_mesh_node { pub index_begin: u32, pub index_count: u32 }
/// rth_pbr_material / rth_vertex_attrs (rt_host.h): PbrMaterial and the rest of GeometryVertex (geometry_import.rs:15-24, 47-55)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rth_pbr_material {
    pub base_color_factor: [f32; 3], pub metallic_factor: f32, pub roughness_factor: f32,
    pub base_color_texarray_id: u32, pub metallic_rough_texarray_id: u32, pub normal_texarray_id: u32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rth_vertex_attrs { pub color: [f32; 4], pub tangent: [f32; 4], pub pbr_buf_id: u32 }
#[repr(C)]
pub struct rtg_perlin { pub ranvec: [[f32; 3]; 256], pub perm_x: [i32; 256], pub perm_y: [i32; 256], pub perm_z: [i32; 256] }

/// rth_image / rth_scene_assets (rt_host.h): what rth_build_scene needs for the scenes that load files — only for a host that
/// lets the library build the reference's scenes instead of flattening its own (tests, benchmarks)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rth_image { pub width: i32, pub height: i32, pub rgba: *const u8 }
#[repr(C)]
pub struct rth_scene_assets {
    pub earthmap: rth_image,
    pub uvgrid01: rth_image,
    pub uvgrid09: rth_image,
    pub teapot: *const rth_geometry,
    pub mesh_grid: i32,
    pub uvgrid03: rth_image,
    pub uvgrid08: rth_image,
    pub uvgrid10: rth_image,
}

#[link(name = "rtiow_hip")]
extern "C" {
    pub fn rt_device_count() -> c_int;
    pub fn rt_create(device_id: c_int, out: *mut *mut rt_ctx) -> c_int;
    pub fn rt_destroy(ctx: *mut rt_ctx);
    pub fn rt_last_error(ctx: *const rt_ctx) -> *const c_char;
    pub fn rt_scene_upload(ctx: *mut rt_ctx, scene: *const rt_scene_desc) -> c_int;
    pub fn rt_scene_query(ctx: *mut rt_ctx, info: *mut rt_scene_info) -> c_int;
    pub fn rt_scene_read_bvh(ctx: *mut rt_ctx, nodes: *mut f32, slots: *mut f32) -> c_int;
    pub fn rt_sampler_upload(ctx: *mut rt_ctx, sampler: *const rt_sampler_desc) -> c_int;
    pub fn rt_render(ctx: *mut rt_ctx, params: *const rt_render_params, out_rgba_sum: *mut f32, stats: *mut rt_stats) -> c_int;
    pub fn rt_resolve(ctx: *mut rt_ctx, rgba_sum: *const f32, n_pixels: i32, spp: i32, out_rgba: *mut f32, location: i32, stream: *mut c_void) -> c_int;
    pub fn rt_resolve_rgb8(ctx: *mut rt_ctx, rgba_sum: *const f32, n_pixels: i32, spp: i32, out_rgb: *mut u8, location: i32, stream: *mut c_void) -> c_int;
    pub fn rt_trace_first_hit(ctx: *mut rt_ctx, rays_soa: *const f32, n: i32, t_min: f32, t_max: f32, prim_id: *mut u32, t: *mut f32,
                              normal3: *mut f32, front_face: *mut u8, uv2: *mut f32) -> c_int;
    pub fn rt_trace_first_hit_noalloc(ctx: *mut rt_ctx, rays_soa: *const f32, n: i32, t_min: f32, t_max: f32, prim_id: *mut u32, t: *mut f32,
                                      normal3: *mut f32, front_face: *mut u8, uv2: *mut f32) -> c_int;
    pub fn rt_last_kernel_stats(ctx: *mut rt_ctx, kernel: i32, out: *mut rt_stats) -> c_int;
    pub fn rt_synchronize(ctx: *mut rt_ctx, stream: *mut c_void) -> c_int;
    pub fn rt_device_info(ctx: *mut rt_ctx, name: *mut c_char, name_len: i32, cu_count: *mut i32, clock_mhz: *mut i32) -> c_int;
    // multi-GPU
    pub fn rt_device_alloc(ctx: *mut rt_ctx, bytes: usize, out_device_ptr: *mut *mut c_void) -> c_int;
    pub fn rt_device_free(ctx: *mut rt_ctx, device_ptr: *mut c_void) -> c_int;
    pub fn rt_device_zero(ctx: *mut rt_ctx, device_ptr: *mut c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn rt_device_read(ctx: *mut rt_ctx, dst_host: *mut c_void, src_device: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn rt_stream_create(ctx: *mut rt_ctx, out_stream: *mut *mut c_void) -> c_int;
    pub fn rt_stream_destroy(ctx: *mut rt_ctx, stream: *mut c_void) -> c_int;
    pub fn rt_stream_wait(ctx: *mut rt_ctx, stream: *mut c_void, after: *mut c_void) -> c_int;
    pub fn rt_comm_unique_id(id: *mut c_char) -> c_int;
    pub fn rt_comm_create(ctx: *mut rt_ctx, id: *const c_char, rank: c_int, world: c_int, out: *mut *mut rt_comm) -> c_int;
    pub fn rt_comm_reduce_to_root(comm: *mut rt_comm, rgba_sum_device: *mut f32, n_floats: i64, root: c_int, stream: *mut c_void) -> c_int;
    pub fn rt_comm_destroy(comm: *mut rt_comm);
    pub fn rt_group_create(device_ids: *const c_int, n: c_int, out: *mut *mut rt_group) -> c_int;
    pub fn rt_group_size(g: *const rt_group) -> c_int;
    pub fn rt_group_ctx(g: *mut rt_group, i: c_int) -> *mut rt_ctx;
    pub fn rt_group_reduce_to_root(g: *mut rt_group, rgba_sums: *const *mut f32, n_floats: i64, root: c_int, streams: *const *mut c_void) -> c_int;
    pub fn rt_group_destroy(g: *mut rt_group);
}

#[link(name = "rtiow_host")]
extern "C" {
    pub fn rth_last_error() -> *const c_char;
    pub fn rth_scene_new() -> *mut rth_scene;
    pub fn rth_scene_free(s: *mut rth_scene);
    pub fn rth_scene_graph(s: *mut rth_scene) -> *const rtg_graph;
    // textures, materials (return the index, or a negative rt_status)
    pub fn rth_add_solid_texture(s: *mut rth_scene, r: f32, g: f32, b: f32) -> c_int;
    pub fn rth_add_checker_texture(s: *mut rth_scene, odd_tex: c_int, even_tex: c_int, repeat: f32) -> c_int;
    pub fn rth_add_noise_texture(s: *mut rth_scene, scale: f32, seed: u32) -> c_int;
    pub fn rth_add_noise_texture_table(s: *mut rth_scene, scale: f32, table: *const rtg_perlin) -> c_int;
    pub fn rth_add_image_texture(s: *mut rth_scene, width: c_int, height: c_int, rgba: *const u8) -> c_int;
    pub fn rth_add_lambertian(s: *mut rth_scene, albedo_tex: c_int) -> c_int;
    pub fn rth_add_metal(s: *mut rth_scene, r: f32, g: f32, b: f32, fuzz: f32) -> c_int;
    pub fn rth_add_dielectric(s: *mut rth_scene, ior: f32) -> c_int;
    pub fn rth_add_diffuse_light(s: *mut rth_scene, emit_tex: c_int) -> c_int;
    pub fn rth_add_isotropic(s: *mut rth_scene, albedo_tex: c_int) -> c_int;
    // hittables
    pub fn rth_add_sphere(s: *mut rth_scene, c: *const f32, r: f32, mtl: c_int) -> c_int;
    pub fn rth_add_moving_sphere(s: *mut rth_scene, c0: *const f32, c1: *const f32, t0: f32, t1: f32, r: f32, mtl: c_int) -> c_int;
    pub fn rth_add_rect(s: *mut rth_scene, rtg_type: c_int, a0: f32, a1: f32, b0: f32, b1: f32, k: f32, mtl: c_int) -> c_int;
    pub fn rth_add_block(s: *mut rth_scene, p0: *const f32, p1: *const f32, mtl: c_int) -> c_int;
    pub fn rth_add_flip_face(s: *mut rth_scene, child: c_int) -> c_int;
    pub fn rth_add_translate(s: *mut rth_scene, child: c_int, offset: *const f32) -> c_int;
    pub fn rth_add_rotate_y(s: *mut rth_scene, child: c_int, degrees: f32) -> c_int;
    pub fn rth_add_rotate_y_sincos(s: *mut rth_scene, child: c_int, sin_theta: f32, cos_theta: f32) -> c_int;
    pub fn rth_add_bvh(s: *mut rth_scene, children: *const i32, n: c_int) -> c_int;
    pub fn rth_add_list(s: *mut rth_scene, children: *const i32, n: c_int) -> c_int;
    pub fn rth_add_constant_medium(s: *mut rth_scene, boundary: c_int, albedo_tex: c_int, density: f32) -> c_int;
    pub fn rth_add_triangle_mesh(s: *mut rth_scene, verts: *const rtg_vertex, n_verts: u32, indices: *const u32, n_indices: u32,
                                 nodes: *const rtg_mesh_node, n_nodes: u32, obj2world: *const f32 /* null = already baked */, mtl: c_int) -> c_int;
    pub fn rth_add_transform(s: *mut rth_scene, child: c_int, obj2world: *const f32) -> c_int;
    pub fn rth_add_cylinder(s: *mut rth_scene, radius: f32, z0: f32, z1: f32, phi_max: f32, mtl: c_int) -> c_int;
    pub fn rth_add_cone(s: *mut rth_scene, radius: f32, height: f32, phi_max: f32, mtl: c_int) -> c_int;
    pub fn rth_add_paraboloid(s: *mut rth_scene, radius: f32, zmin: f32, zmax: f32, phi_max: f32, mtl: c_int) -> c_int;
    pub fn rth_add_hyperboloid(s: *mut rth_scene, p1: *const f32, p2: *const f32, phi_max: f32, mtl: c_int) -> c_int;
    pub fn rth_set_roots(s: *mut rth_scene, world_list: c_int, lights_list: c_int) -> c_int;
    // flatten
    pub fn rth_flatten(g: *const rtg_graph, out: *mut *mut rth_flat) -> c_int;
    pub fn rth_flatten_opts(g: *const rtg_graph, flags: u32, out: *mut *mut rth_flat) -> c_int;
    pub fn rth_flat_free(f: *mut rth_flat);
    pub fn rth_flat_desc(f: *const rth_flat, out: *mut rt_scene_desc);
    // optional helpers: the host keeps its own config loader / camera / sampler / image decoder (serde, camera.rs, sampling.rs,
    // the `image` crate); these are the library's restatements of them for hosts that have none
    pub fn rth_config_load(path: *const c_char, out: *mut *mut rth_config) -> c_int;
    pub fn rth_config_parse(text: *const c_char, out: *mut *mut rth_config) -> c_int;
    pub fn rth_config_free(c: *mut rth_config);
    pub fn rth_config_active_scene(c: *const rth_config) -> *const c_char;
    pub fn rth_config_scene_params(c: *const rth_config, scene: *const c_char, out: *mut rth_user_config) -> c_int;
    pub fn rth_image_height(c: *const rth_user_config) -> i32;
    pub fn rth_camera_new(look_from: *const f32, look_at: *const f32, world_up: *const f32, vfov_degrees: f32, aspect: f32, aperture: f32,
                          focus_dist: f32, time0: f32, time1: f32, out: *mut rt_camera);
    pub fn rth_sampler_new(spp: i32, sets: i32, seed: u32, out: *mut *mut rth_sampler) -> c_int;
    pub fn rth_sampler_free(s: *mut rth_sampler);
    pub fn rth_sampler_desc(s: *const rth_sampler, out: *mut rt_sampler_desc);
    pub fn rth_build_scene(s: *mut rth_scene, name: *const c_char, seed: u32, assets: *const rth_scene_assets) -> c_int;
    pub fn rth_geometry_import_glb(path: *const c_char, out: *mut *mut rth_geometry) -> c_int;
    pub fn rth_geometry_free(g: *mut rth_geometry);
    pub fn rth_geometry_view(g: *const rth_geometry, verts: *mut *const rtg_vertex, n_verts: *mut u32, indices: *mut *const u32, n_indices: *mut u32,
                             nodes: *mut *const rtg_mesh_node, n_nodes: *mut u32);
    pub fn rth_geometry_pbr_materials(g: *const rth_geometry, materials: *mut *const rth_pbr_material, n_materials: *mut u32);
    pub fn rth_geometry_vertex_attrs(g: *const rth_geometry, attrs: *mut *const rth_vertex_attrs, n_verts: *mut u32);
    pub fn rth_geometry_material_index(g: *const rth_geometry, name: *const c_char) -> i32;
    pub fn rth_geometry_pbr_images(g: *const rth_geometry, kind: c_int, width: *mut u32, height: *mut u32, n_layers: *mut u32,
                                   rgba_layers: *mut *const *const u8, layer_sizes: *mut *const u32) -> c_int;
    pub fn rth_mat4_mul(a: *const f32, b: *const f32, out: *mut f32);
    pub fn rth_mat4_invert(m: *const f32, out: *mut f32) -> c_int;
    pub fn rth_mat4_transpose(m: *const f32, out: *mut f32);
    pub fn rth_color_from_u32(argb: u32, out_rgba: *mut f32);
    pub fn rth_rgba8_to_u32(r: u8, g: u8, b: u8, a: u8) -> u32;
    pub fn rth_rgba8_from_html(text: *const c_char, out_rgba: *mut u8);
    pub fn rth_work_blocks(width: i32, height: i32, block_pixels: i32, out_xyxy: *mut i32, cap: i32) -> i32;
    pub fn rth_image_decode(data: *const u8, size: usize, width: *mut i32, height: *mut i32, rgba: *mut *mut u8) -> c_int;
    pub fn rth_image_decode_file(path: *const c_char, width: *mut i32, height: *mut i32, rgba: *mut *mut u8) -> c_int;
    pub fn rth_image_free(rgba: *mut u8);
    pub fn rth_write_png_rgb8(path: *const c_char, width: i32, height: i32, rgb: *const u8) -> c_int;
    pub fn rth_checkpoint_save(path: *const c_char, width: i32, height: i32, samples_per_pixel: i32, spp_done: i32, seed: u32, rgba_sum: *const f32) -> c_int;
    pub fn rth_checkpoint_load(path: *const c_char, width: i32, height: i32, samples_per_pixel: i32, seed: u32, rgba_sum: *mut f32, spp_done: *mut i32) -> c_int;
}

/// rt_status -> Result with the library's message
pub fn check(ctx: *const rt_ctx, status: c_int) -> Result<(), String> {
    if status == RT_OK { return Ok(()); }
    let msg = unsafe { std::ffi::CStr::from_ptr(rt_last_error(ctx)) }.to_string_lossy().into_owned();
    let host = unsafe { std::ffi::CStr::from_ptr(rth_last_error()) }.to_string_lossy().into_owned();
    Err(format!("rt_status {}: {} {}", status, msg, host))
}
