"""Host side of the backend: thin Python handles over librtiow_host.so.

Names follow the reference (raytracer/src/main.rs): RaytracerConfig / RaytracerParams
(:1135-1216), Camera (camera.rs:25-80), MultiJitteredSampler (sampling.rs:255), the
scene_*() builders (:106-1029).  All the work is in C++ (csrc/rt_host_*.cpp); Python only
carries pointers.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from ._ffi import RtError

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA_DIR = os.path.join(REPO_ROOT, "data")
DEFAULT_CONFIG = os.path.join(DATA_DIR, "config", "raytracer.config.ron")
PROJECTIONS = ["Perspective", "Orthographic", "FishEye", "SphericalPanoramic"]


def _check(status):
    if status < 0:
        raise RtError(status, _ffi.host_lib().rth_last_error().decode())
    return status


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class RaytracerParams:
    """RaytracerParams (main.rs:1157-1203): user config + derived image_height."""

    def __init__(self, user):
        self.user = user
        for name, _ in _ffi.rth_user_config._fields_:
            v = getattr(user, name)
            setattr(self, name, list(v) if hasattr(v, "__len__") else v)
        self.image_height = _ffi.host_lib().rth_image_height(C.byref(user))
        self.projection_name = PROJECTIONS[user.projection]
        # main.rs:1199-1200: f32::to_radians = x * (PI_f32 / 180_f32), one f32 multiply
        k = np.float32(np.float32(np.pi) / np.float32(180.0))
        self.psi_max_radians = float(np.float32(user.psi_max) * k)
        self.lambda_max_radians = float(np.float32(user.lambda_max) * k)

    def replace(self, **kw):
        u = _ffi.rth_user_config()
        C.memmove(C.byref(u), C.byref(self.user), C.sizeof(u))
        for k, v in kw.items():
            if k in ("look_from", "look_at", "world_up", "background"):
                setattr(u, k, _f3(v))
            else:
                setattr(u, k, v)
        return RaytracerParams(u)

    def camera(self):
        return Camera.new(self.look_from, self.look_at, self.world_up, self.vertical_fov, self.aspect_ratio,
                          self.aperture, self.focus_dist, 0.0, 1.0)  # main.rs:1288-1298


class RaytracerConfig:
    """RaytracerConfig (main.rs:1211-1216) loaded from the RON file (main.rs:1241-1246)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load(cls, path=DEFAULT_CONFIG):
        h = C.c_void_p()
        _check(_ffi.host_lib().rth_config_load(path.encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def parse(cls, text):
        h = C.c_void_p()
        _check(_ffi.host_lib().rth_config_parse(text.encode(), C.byref(h)))
        return cls(h)

    @property
    def active_scene(self):
        return _ffi.host_lib().rth_config_active_scene(self._h).decode()

    def scene_params(self, scene=None):
        u = _ffi.rth_user_config()
        _check(_ffi.host_lib().rth_config_scene_params(self._h, scene.encode() if scene else None, C.byref(u)))
        return RaytracerParams(u)

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.host_lib().rth_config_free(self._h)
            self._h = None


class Camera:
    @staticmethod
    def new(look_from, look_at, world_up, vertical_fov, aspect_ratio, aperture, focus_dist, time0=0.0, time1=1.0):
        cam = _ffi.rt_camera()
        _ffi.host_lib().rth_camera_new(_f3(look_from), _f3(look_at), _f3(world_up), vertical_fov, aspect_ratio,
                                       aperture, focus_dist, time0, time1, C.byref(cam))
        return cam


class MultiJitteredSampler:
    """SamplerBase<MultiJitteredSamplingStrategy>::new(spp, None) (sampling.rs:30-135)."""

    def __init__(self, spp, sets=0, seed=0x5EED):
        self._h = C.c_void_p()
        _check(_ffi.host_lib().rth_sampler_new(int(spp), int(sets), int(seed), C.byref(self._h)))
        self.desc = _ffi.rt_sampler_desc()
        _ffi.host_lib().rth_sampler_desc(self._h, C.byref(self.desc))
        self.spp = int(spp)

    def arrays(self):
        n = self.desc.sets * self.desc.samples_in_set
        return (np.ctypeslib.as_array(self.desc.samples, (n, 2)).copy(),
                np.ctypeslib.as_array(self.desc.disk_samples, (n, 2)).copy(),
                np.ctypeslib.as_array(self.desc.shuffled_indices, (n,)).copy())

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.host_lib().rth_sampler_free(self._h)
            self._h = None


class Geometry:
    """ImportedGeometry::import_from_file (geometry_import.rs:445-507), geometry only."""

    def __init__(self, path):
        self._h = C.c_void_p()
        _check(_ffi.host_lib().rth_geometry_import_glb(path.encode(), C.byref(self._h)))

    def view(self):
        v, nv = C.POINTER(_ffi.rtg_vertex)(), C.c_uint32()
        ix, ni = _ffi.c_u32_p(), C.c_uint32()
        nd, nn = C.POINTER(_ffi.rtg_mesh_node)(), C.c_uint32()
        _ffi.host_lib().rth_geometry_view(self._h, C.byref(v), C.byref(nv), C.byref(ix), C.byref(ni), C.byref(nd), C.byref(nn))
        verts = np.ctypeslib.as_array(C.cast(v, _ffi.c_float_p), (nv.value, 8)).copy()
        idx = np.ctypeslib.as_array(ix, (ni.value,)).copy()
        nodes = np.ctypeslib.as_array(C.cast(nd, _ffi.c_u32_p), (nn.value, 2)).copy()
        return verts, idx, nodes

    def pbr_materials(self):
        """ImportedGeometry::pbr_materials (geometry_import.rs:96-98): structured array, one PbrMaterial per glTF material."""
        m, n = C.POINTER(_ffi.rth_pbr_material)(), C.c_uint32()
        _ffi.host_lib().rth_geometry_pbr_materials(self._h, C.byref(m), C.byref(n))
        dt = np.dtype([("base_color_factor", "<f4", 3), ("metallic_factor", "<f4"), ("roughness_factor", "<f4"),
                       ("base_color_texarray_id", "<u4"), ("metallic_rough_texarray_id", "<u4"), ("normal_texarray_id", "<u4")])
        if n.value == 0:
            return np.zeros(0, dt)
        return np.frombuffer(C.string_at(m, n.value * C.sizeof(_ffi.rth_pbr_material)), dtype=dt).copy()

    def vertex_attrs(self):
        """colour, tangent and pbr_buf_id of every vertex (geometry_import.rs:47-68), parallel to view()'s vertices."""
        a, n = C.POINTER(_ffi.rth_vertex_attrs)(), C.c_uint32()
        _ffi.host_lib().rth_geometry_vertex_attrs(self._h, C.byref(a), C.byref(n))
        dt = np.dtype([("color", "<f4", 4), ("tangent", "<f4", 4), ("pbr_buf_id", "<u4")])
        if n.value == 0:
            return np.zeros(0, dt)
        return np.frombuffer(C.string_at(a, n.value * C.sizeof(_ffi.rth_vertex_attrs)), dtype=dt).copy()

    def material_index(self, name):
        return int(_ffi.host_lib().rth_geometry_material_index(self._h, name.encode()))

    def pbr_images(self, kind):
        """pbr_base_color_images / pbr_metallic_roughness_images / pbr_normal_images (kind 0 / 1 / 2): (width, height, [RGBA8 layers])."""
        w, h, n = C.c_uint32(), C.c_uint32(), C.c_uint32()
        layers, sizes = C.POINTER(_ffi.c_u8_p)(), _ffi.c_u32_p()
        _check(_ffi.host_lib().rth_geometry_pbr_images(self._h, kind, C.byref(w), C.byref(h), C.byref(n), C.byref(layers), C.byref(sizes)))
        out = [np.ctypeslib.as_array(layers[i], (sizes[2 * i + 1], sizes[2 * i], 4)).copy() for i in range(n.value)]
        return w.value, h.value, out

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.host_lib().rth_geometry_free(self._h)
            self._h = None


def _take_image(status, w, h, px):
    _check(status)
    try:
        return np.ctypeslib.as_array(px, (h.value, w.value, 4)).copy()
    finally:
        _ffi.host_lib().rth_image_free(px)


def load_image_rgba(path):
    """ImageTexture::new (image_texture.rs:20-38): decode + into_rgba8, H x W x 4 uint8.  The host library's own
    decoder (csrc/rt_host_image.cpp: baseline JPEG restating jpeg-decoder 0.3.0, PNG); no PIL on the asset path."""
    w, h, px = C.c_int32(), C.c_int32(), _ffi.c_u8_p()
    return _take_image(_ffi.host_lib().rth_image_decode_file(os.fspath(path).encode(), C.byref(w), C.byref(h), C.byref(px)), w, h, px)


def decode_image_rgba(data):
    """The same for a file already in memory (e.g. an image embedded in a GLB, geometry_import.rs:461-488)."""
    w, h, px = C.c_int32(), C.c_int32(), _ffi.c_u8_p()
    data = bytes(data)
    return _take_image(_ffi.host_lib().rth_image_decode(data, len(data), C.byref(w), C.byref(h), C.byref(px)), w, h, px)


class Assets:
    """Files the reference's scene builders open by path (main.rs:778, 836-925, 975, 1001, 1012, 1037)."""

    def __init__(self, data_dir=DATA_DIR, mesh_grid=1, need=("earthmap", "uvgrid01", "uvgrid09", "teapot")):
        self.c = _ffi.rth_scene_assets()
        self._keep = []
        files = {"earthmap": "textures/misc/earthmap.jpg", "uvgrid01": "textures/uv_grids/ash_uvgrid01.jpg",
                 "uvgrid09": "textures/uv_grids/ash_uvgrid09.jpg", "uvgrid03": "textures/uv_grids/ash_uvgrid03.jpg",
                 "uvgrid08": "textures/uv_grids/ash_uvgrid08.jpg", "uvgrid10": "textures/uv_grids/ash_uvgrid10.jpg"}
        for k, rel in files.items():
            if k not in need:
                continue
            a = load_image_rgba(os.path.join(data_dir, rel))
            self._keep.append(a)
            img = getattr(self.c, k)
            img.width, img.height = a.shape[1], a.shape[0]
            img.rgba = a.ctypes.data_as(_ffi.c_u8_p)
        self.teapot = None
        if "teapot" in need:
            self.teapot = Geometry(os.path.join(data_dir, "models/teapot/teapot.glb"))
            self.c.teapot = self.teapot._h
        self.c.mesh_grid = mesh_grid


class FlatScene:
    """Result of the flatten step: host arrays in the layout of include/rt_api.h."""

    def __init__(self, graph_ptr, owner, device_bvh=False):
        self._owner = owner
        self._h = C.c_void_p()
        _check(_ffi.host_lib().rth_flatten_opts(graph_ptr, 1 if device_bvh else 0, C.byref(self._h)))
        self.desc = _ffi.rt_scene_desc()
        _ffi.host_lib().rth_flat_desc(self._h, C.byref(self.desc))

    def slots(self):
        return np.ctypeslib.as_array(self.desc.slots, (self.desc.n_slots, 16)).copy() if self.desc.n_slots else np.zeros((0, 16), np.float32)

    def xforms(self):
        return np.ctypeslib.as_array(self.desc.xforms, (self.desc.n_xforms, 8)).copy() if self.desc.n_xforms else np.zeros((0, 8), np.float32)

    def media(self):
        return np.ctypeslib.as_array(self.desc.media, (self.desc.n_media, 8)).copy() if self.desc.n_media else np.zeros((0, 8), np.float32)

    def nodes(self):
        return np.ctypeslib.as_array(self.desc.bvh_nodes, (self.desc.n_bvh_nodes, 16)).copy() if self.desc.n_bvh_nodes else np.zeros((0, 16), np.float32)

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.host_lib().rth_flat_free(self._h)
            self._h = None


class Scene:
    """A `(world, lights)` pair as returned by the reference's scene_*() functions."""

    SUPPORTED = ("RandomWorld", "TwoSpheres", "TexturedSpheres", "PerlinSpheres", "SimpleLight", "CornellBox", "Chapter2Final",
                 "MeshTest", "GeometricPrimitives", "Blocks")          # every `Scene` variant of main.rs:91-103

    def __init__(self):
        self._h = C.c_void_p(_ffi.host_lib().rth_scene_new())
        self._assets = None

    @classmethod
    def build(cls, name, seed=0x5EED, assets=None):
        s = cls()
        s._assets = assets
        _check(_ffi.host_lib().rth_build_scene(s._h, name.encode(), int(seed), C.byref(assets.c) if assets else None))
        return s

    # constructors of the reference types, for hand-made scenes (tests)
    def solid(self, r, g, b): return _check(_ffi.host_lib().rth_add_solid_texture(self._h, r, g, b))
    def checker(self, odd, even, repeat): return _check(_ffi.host_lib().rth_add_checker_texture(self._h, odd, even, repeat))
    def noise(self, scale, seed=0x5EED): return _check(_ffi.host_lib().rth_add_noise_texture(self._h, scale, seed))

    def noise_table(self, scale, ranvec, perm_x, perm_y, perm_z):
        tab = np.zeros(256 * 3 + 3 * 256, np.uint32)
        tab[:768] = np.ascontiguousarray(ranvec, np.float32).reshape(768).view(np.uint32)
        tab[768:1024], tab[1024:1280], tab[1280:] = perm_x, perm_y, perm_z
        return _check(_ffi.host_lib().rth_add_noise_texture_table(self._h, scale, tab.ctypes.data))

    def image(self, rgba):
        if rgba is None:
            return _check(_ffi.host_lib().rth_add_image_texture(self._h, 0, 0, None))
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        return _check(_ffi.host_lib().rth_add_image_texture(self._h, rgba.shape[1], rgba.shape[0], rgba.ctypes.data_as(_ffi.c_u8_p)))

    def lambertian(self, tex): return _check(_ffi.host_lib().rth_add_lambertian(self._h, tex))
    def metal(self, r, g, b, fuzz): return _check(_ffi.host_lib().rth_add_metal(self._h, r, g, b, fuzz))
    def dielectric(self, ior): return _check(_ffi.host_lib().rth_add_dielectric(self._h, ior))
    def diffuse_light(self, tex): return _check(_ffi.host_lib().rth_add_diffuse_light(self._h, tex))
    def sphere(self, c, r, mtl): return _check(_ffi.host_lib().rth_add_sphere(self._h, _f3(c), r, mtl))
    def moving_sphere(self, c0, c1, t0, t1, r, mtl): return _check(_ffi.host_lib().rth_add_moving_sphere(self._h, _f3(c0), _f3(c1), t0, t1, r, mtl))
    def xy_rect(self, x0, x1, y0, y1, k, mtl): return _check(_ffi.host_lib().rth_add_rect(self._h, _ffi.RTG_XY_RECT, x0, x1, y0, y1, k, mtl))
    def xz_rect(self, x0, x1, z0, z1, k, mtl): return _check(_ffi.host_lib().rth_add_rect(self._h, _ffi.RTG_XZ_RECT, x0, x1, z0, z1, k, mtl))
    def yz_rect(self, y0, y1, z0, z1, k, mtl): return _check(_ffi.host_lib().rth_add_rect(self._h, _ffi.RTG_YZ_RECT, y0, y1, z0, z1, k, mtl))
    def block(self, p0, p1, mtl): return _check(_ffi.host_lib().rth_add_block(self._h, _f3(p0), _f3(p1), mtl))
    def flip_face(self, child): return _check(_ffi.host_lib().rth_add_flip_face(self._h, child))
    def translate(self, child, offset): return _check(_ffi.host_lib().rth_add_translate(self._h, child, _f3(offset)))
    def rotate_y(self, child, degrees): return _check(_ffi.host_lib().rth_add_rotate_y(self._h, child, degrees))
    def rotate_y_sincos(self, child, sin_theta, cos_theta):
        """RotateY with the sin / cos its owner already holds (transform.rs:57-62)."""
        return _check(_ffi.host_lib().rth_add_rotate_y_sincos(self._h, child, sin_theta, cos_theta))
    def transform(self, child, obj2world):
        """Transform::new(obj2world, child) (transform.rs:162-175); obj2world = 4x4 row-major, affine."""
        m = (C.c_float * 16)(*[float(x) for x in np.asarray(obj2world, np.float32).reshape(16)])
        return _check(_ffi.host_lib().rth_add_transform(self._h, child, m))

    def cylinder(self, radius, z0, z1, phi_max, mtl): return _check(_ffi.host_lib().rth_add_cylinder(self._h, radius, z0, z1, phi_max, mtl))
    def cone(self, radius, height, phi_max, mtl): return _check(_ffi.host_lib().rth_add_cone(self._h, radius, height, phi_max, mtl))
    def paraboloid(self, radius, zmin, zmax, phi_max, mtl): return _check(_ffi.host_lib().rth_add_paraboloid(self._h, radius, zmin, zmax, phi_max, mtl))
    def hyperboloid(self, p1, p2, phi_max, mtl): return _check(_ffi.host_lib().rth_add_hyperboloid(self._h, _f3(p1), _f3(p2), phi_max, mtl))

    def constant_medium(self, boundary, tex, density): return _check(_ffi.host_lib().rth_add_constant_medium(self._h, boundary, tex, density))

    def _ids(self, children):
        arr = (C.c_int32 * len(children))(*children)
        return arr, len(children)

    def bvh(self, children): return _check(_ffi.host_lib().rth_add_bvh(self._h, *self._ids(children)))
    def hittable_list(self, children): return _check(_ffi.host_lib().rth_add_list(self._h, *self._ids(children)))

    def triangle_mesh(self, verts, indices, nodes, obj2world, mtl):
        verts = np.ascontiguousarray(verts, np.float32)
        indices = np.ascontiguousarray(indices, np.uint32)
        nodes = np.ascontiguousarray(nodes, np.uint32)
        # obj2world None: the vertices are world-space already (TriangleMesh::vertices after triangle_mesh.rs:53-72)
        m = None if obj2world is None else C.byref((C.c_float * 16)(*[float(x) for x in np.asarray(obj2world, np.float32).reshape(16)]))
        return _check(_ffi.host_lib().rth_add_triangle_mesh(
            self._h, C.cast(verts.ctypes.data, C.POINTER(_ffi.rtg_vertex)), verts.shape[0],
            indices.ctypes.data_as(_ffi.c_u32_p), indices.shape[0],
            C.cast(nodes.ctypes.data, C.POINTER(_ffi.rtg_mesh_node)), nodes.shape[0], m, mtl))

    def set_roots(self, world, lights): _check(_ffi.host_lib().rth_set_roots(self._h, world, lights))

    def graph(self):
        return _ffi.host_lib().rth_scene_graph(self._h)

    def flatten(self, device_bvh=False):
        """device_bvh: leave the BVH to rt_scene_upload, which then builds it on the GPU (rt_bvh_build.hip)."""
        return FlatScene(self.graph(), self, device_bvh)

    def __del__(self):
        if getattr(self, "_h", None):
            _ffi.host_lib().rth_scene_free(self._h)
            self._h = None


def write_png(path, rgb8):
    """The screenshot file of main.rs:1543-1556: H x W x 3 uint8 (HipBackend.resolve_rgb8) as a PNG."""
    a = np.ascontiguousarray(rgb8, np.uint8)
    assert a.ndim == 3 and a.shape[2] == 3
    _check(_ffi.host_lib().rth_write_png_rgb8(path.encode(), a.shape[1], a.shape[0], a.ctypes.data_as(_ffi.c_u8_p)))


def checkpoint_save(path, sums, samples_per_pixel, spp_done, seed=0x5EED):
    """Resumable accumulation: the H x W x 4 f32 sums of rt_render and how many samples they hold."""
    a = np.ascontiguousarray(sums, np.float32)
    _check(_ffi.host_lib().rth_checkpoint_save(path.encode(), a.shape[1], a.shape[0], samples_per_pixel, spp_done, seed,
                                               a.ctypes.data_as(_ffi.c_float_p)))


def checkpoint_load(path, width, height, samples_per_pixel, seed=0x5EED):
    """-> (sums, spp_done); raises RtError if the file is damaged or belongs to another frame."""
    a = np.zeros((height, width, 4), np.float32)
    done = C.c_int32()
    _check(_ffi.host_lib().rth_checkpoint_load(path.encode(), width, height, samples_per_pixel, seed, a.ctypes.data_as(_ffi.c_float_p),
                                               C.byref(done)))
    return a, done.value


def work_blocks(width, height, block_pixels):
    """WorkBlock list of main.rs:1265-1282 as an (n,4) array of x0,x1,y0,y1."""
    L = _ffi.host_lib()
    n = L.rth_work_blocks(width, height, block_pixels, None, 0)
    out = np.zeros((n, 4), np.int32)
    L.rth_work_blocks(width, height, block_pixels, out.ctypes.data_as(C.POINTER(C.c_int32)), n)
    return out
