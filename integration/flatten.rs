//! flatten.rs — `raytracer/src/flatten.rs`: the object tree of `(world, lights)` handed to librtiow_host.so.
//!
//! UNVERIFIED SOURCE (no rustc in the build image; see gpu_ffi.rs).  The C-ABI calls made here are the ones
//! integration/c_host.c and the ctypes tests make; tests/test_host.py::test_constructors_that_take_what_the_reference_objects_hold
//! covers the two constructor forms only a Rust host needs (RotateY by sin/cos, a mesh that is already baked).
//!
//! `Hittable`, `Material` and `Texture` have no introspection (hittable.rs:44-54, material.rs:18-38, texture.rs:3-5), so each
//! trait gets one more method with a default that refuses:
//!
//! ```ignore
//! // hittable.rs:44     fn flatten(&self, b: &mut SceneBuilder) -> FlatResult { Err(FlatError::Unsupported(std::any::type_name::<Self>())) }
//! // material.rs:18     fn flatten(&self, b: &mut SceneBuilder) -> FlatResult { Err(..) }
//! // texture.rs:3       fn flatten(&self, b: &mut SceneBuilder) -> FlatResult { Err(..) }
//! ```
//!
//! Most fields the methods read are private to their modules, so every `fn flatten` below goes INTO the existing
//! `impl Hittable for X` / `impl Material for X` / `impl Texture for X` block of the file named above it; they are
//! collected here so the patch can be read in one place.  An object the backend does not cover keeps the default and the
//! host falls back to its own thread pool (render_gpu.rs).
use crate::gpu_ffi::*;
use std::collections::HashMap;
use std::os::raw::c_int;
use std::sync::Arc;

#[derive(Debug)]
pub enum FlatError {
    Unsupported(&'static str), // this object has no flatten (or the library refused its nesting)
    Library(c_int, String),    // a negative rt_status from rth_*
}
pub type FlatResult = Result<c_int, FlatError>;

/// Owns the rth_scene while the tree is walked.  Materials and textures are cached by `Arc` address so an object shared by a
/// thousand spheres stays one record (scene_random_world shares nothing, scene_cornell_box shares `white` five times).
pub struct SceneBuilder {
    pub scene: *mut rth_scene,
    materials: HashMap<usize, c_int>,
    textures: HashMap<usize, c_int>,
}

impl SceneBuilder {
    pub fn new() -> Self {
        SceneBuilder { scene: unsafe { rth_scene_new() }, materials: HashMap::new(), textures: HashMap::new() }
    }

    /// negative return of an rth_add_* call -> Err with the library's text
    pub fn idx(&self, r: c_int) -> FlatResult {
        if r >= 0 {
            return Ok(r);
        }
        let msg = unsafe { std::ffi::CStr::from_ptr(rth_last_error()) }.to_string_lossy().into_owned();
        if r == RT_E_UNSUPPORTED_PRIMITIVE { Err(FlatError::Unsupported("nesting refused by the library")) } else { Err(FlatError::Library(r, msg)) }
    }

    pub fn material(&mut self, m: &Arc<dyn crate::material::Material>) -> FlatResult {
        let key = Arc::as_ptr(m) as *const () as usize;
        if let Some(&i) = self.materials.get(&key) {
            return Ok(i);
        }
        let i = m.flatten(self)?;
        self.materials.insert(key, i);
        Ok(i)
    }

    pub fn texture(&mut self, t: &Arc<dyn crate::texture::Texture>) -> FlatResult {
        let key = Arc::as_ptr(t) as *const () as usize;
        if let Some(&i) = self.textures.get(&key) {
            return Ok(i);
        }
        let i = t.flatten(self)?;
        self.textures.insert(key, i);
        Ok(i)
    }

    fn list(&mut self, objs: &[Arc<dyn crate::hittable::Hittable>], bvh: bool) -> FlatResult {
        let ids = objs.iter().map(|o| o.flatten(self)).collect::<Result<Vec<c_int>, _>>()?;
        let r = unsafe { if bvh { rth_add_bvh(self.scene, ids.as_ptr(), ids.len() as c_int) } else { rth_add_list(self.scene, ids.as_ptr(), ids.len() as c_int) } };
        self.idx(r)
    }

    /// `(world, lights)` as the scene_*() functions return them (main.rs:1301-1312) -> the descriptor rt_scene_upload takes.
    /// `device_bvh`: leave the tree to the GPU (rt_bvh_build.hip) — for scenes that are rebuilt often.
    pub fn finish(mut self, world: &crate::hittable_list::HittableList, lights: &crate::hittable_list::HittableList, device_bvh: bool) -> Result<FlatScene, FlatError> {
        let w = self.list(world.objects(), false)?;
        let l = self.list(lights.objects(), false)?;
        let r = unsafe { rth_set_roots(self.scene, w, l) };
        self.idx(r)?;
        let mut flat: *mut rth_flat = std::ptr::null_mut();
        let r = unsafe { rth_flatten_opts(rth_scene_graph(self.scene), if device_bvh { RTH_FLATTEN_DEVICE_BVH } else { 0 }, &mut flat) };
        self.idx(r)?;
        let mut desc = std::mem::MaybeUninit::<rt_scene_desc>::zeroed();
        unsafe { rth_flat_desc(flat, desc.as_mut_ptr()) };
        let scene = std::mem::replace(&mut self.scene, std::ptr::null_mut());
        Ok(FlatScene { scene, flat, desc: unsafe { desc.assume_init() } })
    }
}

impl Drop for SceneBuilder {
    fn drop(&mut self) {
        if !self.scene.is_null() {
            unsafe { rth_scene_free(self.scene) }
        }
    }
}

/// The flat arrays (owned by the library) and the descriptor that points into them; keep it until rt_scene_upload returned.
pub struct FlatScene {
    scene: *mut rth_scene,
    flat: *mut rth_flat,
    pub desc: rt_scene_desc,
}
impl Drop for FlatScene {
    fn drop(&mut self) {
        unsafe {
            rth_flat_free(self.flat);
            rth_scene_free(self.scene);
        }
    }
}

// ------------------------------------------------------------------------------------------------ hittables
// hittable_list.rs — also needs `pub fn objects(&self) -> &[Arc<dyn Hittable>] { &self.objects }` (the field is private, :8)
impl crate::hittable_list::HittableList {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult { b.list(&self.objects, false) }
}

// objects/sphere.rs:10-14
impl crate::objects::sphere::Sphere {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let m = b.material(&self.mtl)?;
        b.idx(unsafe { rth_add_sphere(b.scene, self.center.as_ptr(), self.radius, m) })
    }
}

// objects/sphere.rs (MovingSphere: center0, center1, time0, time1, radius, mtl)
impl crate::objects::sphere::MovingSphere {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let m = b.material(&self.mtl)?;
        b.idx(unsafe { rth_add_moving_sphere(b.scene, self.center0.as_ptr(), self.center1.as_ptr(), self.time0, self.time1, self.radius, m) })
    }
}

// rectangles.rs:12-19, 82-89, 152-159 — RTG_XY_RECT = 2, RTG_XZ_RECT = 3, RTG_YZ_RECT = 4 (rt_scene_graph.h)
impl crate::rectangles::XYRect {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let m = b.material(&self.mtl)?;
        b.idx(unsafe { rth_add_rect(b.scene, 2, self.x0, self.x1, self.y0, self.y1, self.k, m) })
    }
}
impl crate::rectangles::XZRect {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let m = b.material(&self.mtl)?;
        b.idx(unsafe { rth_add_rect(b.scene, 3, self.x0, self.x1, self.z0, self.z1, self.k, m) })
    }
}
impl crate::rectangles::YZRect {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let m = b.material(&self.mtl)?;
        b.idx(unsafe { rth_add_rect(b.scene, 4, self.y0, self.y1, self.z0, self.z1, self.k, m) })
    }
}

// block.rs:11-15 — the six sides all carry Block::new's material (block.rs:22-107), so the first side's is the block's.
// The library re-creates the six rectangles in the reference's order (rt_host_flatten.cpp), which fixes the primitive ids.
impl crate::block::Block {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let m = self.sides.objects()[0].material_for_flatten().ok_or(FlatError::Unsupported("Block side"))?; // XYRect::mtl
        let m = b.material(&m)?;
        b.idx(unsafe { rth_add_block(b.scene, self.box_min.as_ptr(), self.box_max.as_ptr(), m) })
    }
}

// flip_face.rs:5-7
impl crate::flip_face::FlipFace {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let c = self.obj.flatten(b)?;
        b.idx(unsafe { rth_add_flip_face(b.scene, c) })
    }
}

// transform.rs:9-12
impl crate::transform::Translate {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let c = self.obj.flatten(b)?;
        b.idx(unsafe { rth_add_translate(b.scene, c, self.offset.as_ptr()) })
    }
}

// transform.rs:57-62 — the struct holds sin and cos, not the angle: hand those over so the device rotates by the host's bits
impl crate::transform::RotateY {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let c = self.obj.flatten(b)?;
        b.idx(unsafe { rth_add_rotate_y_sincos(b.scene, c, self.sin_theta, self.cos_theta) })
    }
}

// transform.rs:155-160 — Mat4 is #[repr(C)] row-major (math/src/mat4.rs:9-41); the library inverts it with the reference's
// own det / adjoint formulas (mat4.rs:536-617), bit for bit what Transform::new computed
impl crate::transform::Transform {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let c = self.obj.flatten(b)?;
        b.idx(unsafe { rth_add_transform(b.scene, c, self.obj2world.as_ptr()) })
    }
}

// bvh.rs:9-13 — the backend builds its own tree with list semantics, so a BvhNode dissolves into the leaves of its subtree,
// in left-to-right order.  `collect_leaves` is a second trait method with the default `out.push(self_arc)`; BvhNode overrides
// it to recurse into `left` and `right` (a leaf BvhNode holds the same Arc twice, bvh.rs:31-33: push it once).
impl crate::bvh::BvhNode {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let mut leaves: Vec<Arc<dyn crate::hittable::Hittable>> = Vec::new();
        self.collect_leaves(&mut leaves);
        b.list(&leaves, true)
    }
    fn collect_leaves(&self, out: &mut Vec<Arc<dyn crate::hittable::Hittable>>) {
        self.left.collect_leaves_arc(&self.left, out);
        if !Arc::ptr_eq(&self.left, &self.right) {
            self.right.collect_leaves_arc(&self.right, out);
        }
    }
}

// constant_medium.rs:13-17 — the library wants the albedo TEXTURE (ConstantMedium::new wraps it in an Isotropic, :23) and the
// density; neg_inv_density = -1/density is exact to invert only for powers of two, so keep `density` as a field next to it
// (one more f32 in the struct) instead of recomputing it.
impl crate::constant_medium::ConstantMedium {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let boundary = self.boundary.flatten(b)?; // spheres only today: anything else -> RT_E_UNSUPPORTED_PRIMITIVE
        let tex = b.texture(self.phase_function.albedo_for_flatten().ok_or(FlatError::Unsupported("phase function"))?)?;
        b.idx(unsafe { rth_add_constant_medium(b.scene, boundary, tex, self.density) })
    }
}

// triangle_mesh.rs:15-23 — `vertices` are world-space already (TriangleMesh::new bakes them, :53-72): pass them with a null
// obj2world so nothing is transformed or normalised twice.  GeometryVertex (geometry_import.rs:48-55) is wider than
// rtg_vertex, so copy pos / normal / uv; every node's index run is appended to one index array.
impl crate::triangle_mesh::TriangleMesh {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let m = b.material(&self.mtl)?;
        let verts: Vec<rtg_vertex> = self.vertices.iter()
            .map(|v| rtg_vertex { pos: [v.pos.x, v.pos.y, v.pos.z], normal: [v.normal.x, v.normal.y, v.normal.z], uv: [v.uv.x, v.uv.y] })
            .collect();
        let mut indices: Vec<u32> = Vec::new();
        let mut nodes: Vec<rtg_mesh_node> = Vec::new();
        for n in &self.nodes {
            nodes.push(rtg_mesh_node { index_begin: indices.len() as u32, index_count: n.indices.len() as u32 });
            indices.extend_from_slice(&n.indices);
        }
        b.idx(unsafe {
            rth_add_triangle_mesh(b.scene, verts.as_ptr(), verts.len() as u32, indices.as_ptr(), indices.len() as u32, nodes.as_ptr(), nodes.len() as u32,
                                  std::ptr::null(), m)
        })
    }
}

// cylinder.rs:12-19, cone.rs:12-17, paraboloid.rs:10-16, hyperboloid.rs:13-23 (p1, p2 after the constructor's swap, :34: the
// library applies the same swap, which is then a no-op, and derives ah / ch as :36-67 does)
impl crate::cylinder::Cylinder {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let m = b.material(&self.mtl)?;
        b.idx(unsafe { rth_add_cylinder(b.scene, self.radius, self.zmin, self.zmax, self.phi_max, m) })
    }
}
impl crate::cone::Cone {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let m = b.material(&self.mtl)?;
        b.idx(unsafe { rth_add_cone(b.scene, self.radius, self.height, self.phi_max, m) })
    }
}
impl crate::paraboloid::Paraboloid {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let m = b.material(&self.mtl)?;
        b.idx(unsafe { rth_add_paraboloid(b.scene, self.radius, self.zmin, self.zmax, self.phi_max, m) })
    }
}
impl crate::hyperboloid::Hyperboloid {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let m = b.material(&self.mtl)?;
        b.idx(unsafe { rth_add_hyperboloid(b.scene, self.p1.as_ptr(), self.p2.as_ptr(), self.phi_max, m) })
    }
}

// ------------------------------------------------------------------------------------------------ materials
// lambertian.rs:13-15
impl crate::lambertian::Lambertian {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let t = b.texture(&self.albedo)?;
        b.idx(unsafe { rth_add_lambertian(b.scene, t) })
    }
}
// metal.rs:6-9 — the library clamps nothing: Metal::new stores fuzziness as given (metal.rs:12-17)
impl crate::metal::Metal {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        b.idx(unsafe { rth_add_metal(b.scene, self.albedo.r, self.albedo.g, self.albedo.b, self.fuzziness) })
    }
}
// dielectric.rs:6-8
impl crate::dielectric::Dielectric {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult { b.idx(unsafe { rth_add_dielectric(b.scene, self.refraction_index) }) }
}
// diffuse_light.rs:13-15
impl crate::diffuse_light::DiffuseLight {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let t = b.texture(&self.emit)?;
        b.idx(unsafe { rth_add_diffuse_light(b.scene, t) })
    }
}
// isotropic.rs:10-12
impl crate::isotropic::Isotropic {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let t = b.texture(&self.albedo)?;
        b.idx(unsafe { rth_add_isotropic(b.scene, t) })
    }
}

// ------------------------------------------------------------------------------------------------ textures
// solid_color_texture.rs:4-6
impl crate::solid_color_texture::SolidColorTexture {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult { b.idx(unsafe { rth_add_solid_texture(b.scene, self.color.r, self.color.g, self.color.b) }) }
}
// checker_texture.rs:8-12
impl crate::checker_texture::CheckerTexture {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let (odd, even) = (b.texture(&self.odd)?, b.texture(&self.even)?);
        b.idx(unsafe { rth_add_checker_texture(b.scene, odd, even, self.repeat_factor) })
    }
}
// noise_texture.rs:7-10 + perlin.rs:5-10 — the host's own tables (thread_rng, unseeded) travel as they are
impl crate::noise_texture::NoiseTexture {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let p = &self.perlin;
        let mut t = Box::new(rtg_perlin { ranvec: [[0.0; 3]; 256], perm_x: [0; 256], perm_y: [0; 256], perm_z: [0; 256] });
        for i in 0..256 {
            t.ranvec[i] = [p.randfloat[i].x, p.randfloat[i].y, p.randfloat[i].z];
            t.perm_x[i] = p.perm_x[i];
            t.perm_y[i] = p.perm_y[i];
            t.perm_z[i] = p.perm_z[i];
        }
        b.idx(unsafe { rth_add_noise_texture_table(b.scene, self.scale, &*t) })
    }
}
// image_texture.rs:3-8 — the bytes the `image` crate decoded, so host and device sample the same texels; 0 x 0 = blank() (:11-18)
impl crate::image_texture::ImageTexture {
    fn flatten(&self, b: &mut SceneBuilder) -> FlatResult {
        let px = if self.pixels.is_empty() { std::ptr::null() } else { self.pixels.as_ptr() };
        b.idx(unsafe { rth_add_image_texture(b.scene, self.width as c_int, self.height as c_int, px) })
    }
}
