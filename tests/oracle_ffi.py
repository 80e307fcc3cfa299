"""ctypes binding of oracle/liboracle.so — test infrastructure, never imported by the product."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, ROOT)
import rtiow_loader
rt = rtiow_loader.load()
ffi = rt._ffi

_lib = None


class orc_options(C.Structure):
    _fields_ = [("bvh_seed", C.c_uint32), ("mesh_nearest", C.c_int32), ("use_libm", C.c_int32), ("exhaustive", C.c_int32)]


class orc_render_params(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("spp_begin", C.c_int32), ("spp_count", C.c_int32), ("max_ray_depth", C.c_int32),
                ("background", C.c_float * 3), ("seed", C.c_uint32),
                ("tile_x0", C.c_int32), ("tile_y0", C.c_int32), ("tile_x1", C.c_int32), ("tile_y1", C.c_int32),
                ("camera", ffi.rt_camera), ("threads", C.c_int32), ("block_pixels", C.c_int32),
                ("projection", C.c_int32), ("psi_max", C.c_float), ("lambda_max", C.c_float)]


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise ImportError(path + " missing: make -C oracle")
        L = C.CDLL(path)
        vp = C.c_void_p
        fp, u32p, u8p = ffi.c_float_p, ffi.c_u32_p, ffi.c_u8_p
        L.orc_last_error.restype = C.c_char_p
        L.orc_scene_create.argtypes = [C.POINTER(ffi.rtg_graph), C.POINTER(orc_options), C.POINTER(vp)]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_render.argtypes = [vp, C.POINTER(orc_render_params), C.POINTER(ffi.rt_sampler_desc), fp,
                                 C.POINTER(C.c_uint64), u32p]
        L.orc_first_hit.argtypes = [vp, fp, C.c_int32, C.c_float, C.c_float, u32p, fp, fp, u8p, fp]
        L.orc_resolve.argtypes = [fp, C.c_int32, C.c_int32, fp]; L.orc_resolve.restype = None
        L.orc_texture_value.argtypes = [vp, C.c_int32, C.c_float, C.c_float, C.c_float * 3, C.c_float * 3]
        L.orc_aabb_hit.argtypes = [C.c_float * 3, C.c_float * 3, C.c_float * 7, C.c_float, C.c_float]
        L.orc_onb.argtypes = [C.c_float * 3, C.c_float * 9]; L.orc_onb.restype = None
        L.orc_schlick.argtypes = [C.c_float, C.c_float]; L.orc_schlick.restype = C.c_float
        L.orc_refract.argtypes = [C.c_float * 3, C.c_float * 3, C.c_float, C.c_float * 3]; L.orc_refract.restype = None
        L.orc_math.argtypes = [C.c_int32, C.c_float, C.c_float, C.c_int32]; L.orc_math.restype = C.c_float
        L.orc_lights_pdf_value.argtypes = [vp, C.c_float * 3, C.c_float * 3]; L.orc_lights_pdf_value.restype = C.c_float
        L.orc_perlin.argtypes = [C.c_void_p, C.c_float * 3, C.c_int32]; L.orc_perlin.restype = C.c_float
        L.orc_resolve_rgb8.argtypes = [fp, C.c_int32, C.c_int32, u8p]; L.orc_resolve_rgb8.restype = None
        L.orc_set_trace.argtypes = [C.c_int32]; L.orc_set_trace.restype = None
        L.orc_bounce.argtypes = [vp, C.c_float * 7, fp, C.c_int32, C.c_float * 22]
        L.orc_project.argtypes = [C.POINTER(orc_render_params), C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float * 6)]
        L.orc_mat4.argtypes = [C.c_int32, C.c_float * 16, C.c_float * 16]
        _lib = L
    return _lib


class OracleScene:
    def __init__(self, scene, bvh_seed=0x5EED, mesh_nearest=True, use_libm=False, exhaustive=False):
        self._scene = scene  # keeps the graph alive
        self._h = C.c_void_p()
        opt = orc_options(bvh_seed, int(mesh_nearest), int(use_libm), int(exhaustive))
        st = lib().orc_scene_create(scene.graph(), C.byref(opt), C.byref(self._h))
        if st < 0:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())

    def render(self, params, camera, sampler, seed=0x5EED, spp_begin=0, spp_count=None, tile=None, threads=8,
               width=None, height=None, want_pixel_rays=False):
        W = width or params.image_width
        H = height or params.image_height
        p = orc_render_params()
        p.image_width, p.image_height = W, H
        p.samples_per_pixel = sampler.spp
        p.spp_begin = spp_begin
        p.spp_count = sampler.spp - spp_begin if spp_count is None else spp_count
        p.max_ray_depth = params.max_ray_depth
        p.background = (C.c_float * 3)(*params.background)
        p.seed = seed
        p.tile_x0, p.tile_y0, p.tile_x1, p.tile_y1 = tile if tile else (0, 0, W, H)
        p.camera = camera
        p.threads = threads
        p.block_pixels = params.worker_block_pixels
        p.projection, p.psi_max, p.lambda_max = params.projection, params.psi_max_radians, params.lambda_max_radians
        out = np.zeros((H, W, 4), np.float32)
        rays = C.c_uint64(0)
        pr = np.zeros((H, W), np.uint32) if want_pixel_rays else None
        st = lib().orc_render(self._h, C.byref(p), C.byref(sampler.desc), out.ctypes.data_as(ffi.c_float_p), C.byref(rays),
                              pr.ctypes.data_as(ffi.c_u32_p) if pr is not None else None)
        if st < 0:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())
        return (out, rays.value, pr) if want_pixel_rays else (out, rays.value)

    def first_hit(self, rays_soa, t_min=0.001, t_max=float("inf")):
        rays_soa = np.ascontiguousarray(rays_soa, np.float32)
        n = rays_soa.shape[1]
        pid = np.zeros(n, np.uint32); t = np.zeros(n, np.float32); nrm = np.zeros((n, 3), np.float32)
        ff = np.zeros(n, np.uint8); uv = np.zeros((n, 2), np.float32)
        st = lib().orc_first_hit(self._h, rays_soa.ctypes.data_as(ffi.c_float_p), n, t_min, t_max,
                                 pid.ctypes.data_as(ffi.c_u32_p), t.ctypes.data_as(ffi.c_float_p),
                                 nrm.ctypes.data_as(ffi.c_float_p), ff.ctypes.data_as(ffi.c_u8_p), uv.ctypes.data_as(ffi.c_float_p))
        if st < 0:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())
        return pid, t, nrm, ff, uv

    def texture_value(self, tex, u, v, p):
        out = (C.c_float * 3)()
        st = lib().orc_texture_value(self._h, tex, u, v, (C.c_float * 3)(*p), out)
        assert st == 0
        return np.array(list(out), np.float32)

    def bounce(self, o, d, xi, time=0.0):
        """One bounce of ray_color with the given random numbers (orc_bounce): dict of kind, prim, t, p, normal, front_face,
        dir, attenuation, pdf, scattering_pdf, emitted, used — floats as numpy float32."""
        xs = np.ascontiguousarray(xi, np.float32)
        out = (C.c_float * 22)()
        st = lib().orc_bounce(self._h, (C.c_float * 7)(*o, *d, time), xs.ctypes.data_as(ffi.c_float_p), len(xs), out)
        assert st == 0
        v = np.array(list(out), np.float32)
        return {"kind": int(v[0]), "prim": int(v[1]), "t": v[2], "p": v[3:6], "normal": v[6:9], "front_face": int(v[9]), "dir": v[10:13],
                "attenuation": v[13:16], "pdf": v[16], "scattering_pdf": v[17], "emitted": v[18:21], "used": int(v[21])}

    def lights_pdf_value(self, o, v):
        return lib().orc_lights_pdf_value(self._h, (C.c_float * 3)(*o), (C.c_float * 3)(*v))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_destroy(self._h)
            self._h = None


def mat4(op, m):
    """The oracle's own matrix code: op 0 transpose, 1 invert, 2 adjoint (orc_mat4)."""
    out = (C.c_float * 16)()
    st = lib().orc_mat4(op, (C.c_float * 16)(*[float(x) for x in m]), out)
    assert st == 0
    return np.array(list(out), np.float32)


def resolve_rgb8(sum_rgba, spp):
    s = np.ascontiguousarray(sum_rgba, np.float32)
    out = np.zeros(s.shape[:-1] + (3,), np.uint8)
    lib().orc_resolve_rgb8(s.ctypes.data_as(ffi.c_float_p), s.size // 4, spp, out.ctypes.data_as(ffi.c_u8_p))
    return out


def resolve(sum_rgba, spp):
    s = np.ascontiguousarray(sum_rgba, np.float32)
    out = np.zeros_like(s)
    lib().orc_resolve(s.ctypes.data_as(ffi.c_float_p), s.size // 4, spp, out.ctypes.data_as(ffi.c_float_p))
    return out
