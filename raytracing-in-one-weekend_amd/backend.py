"""The HIP render path behind include/rt_api.h (librtiow_hip.so).

`HipBackend` stands where the reference's worker pool stands (raytracer/src/main.rs:1314-1409):
it is given the camera, `world`/`lights` (flattened) and the sampler table, and fills
`image_pixels`.  There is no CPU path here; without the library or a GPU it raises.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import RtError


class HipBackend:
    def __init__(self, device_id=0):
        self._L = _ffi.hip_lib()
        self._h = C.c_void_p()
        st = self._L.rt_create(int(device_id), C.byref(self._h))
        if st < 0:
            raise RtError(st, self._L.rt_last_error(None).decode())
        self.device_id = int(device_id)
        self._flat = None
        self._sampler = None

    def _check(self, st):
        if st < 0:
            raise RtError(st, self._L.rt_last_error(self._h).decode())

    def device_info(self):
        name = C.create_string_buffer(256)
        cu, mhz = C.c_int32(), C.c_int32()
        self._check(self._L.rt_device_info(self._h, name, 256, C.byref(cu), C.byref(mhz)))
        return {"name": name.value.decode(), "cu_count": cu.value, "clock_mhz": mhz.value}

    def upload_scene(self, flat):
        """flat: host.FlatScene (or anything with a .desc rt_scene_desc)."""
        self._check(self._L.rt_scene_upload(self._h, C.byref(flat.desc)))
        self._flat = flat

    def scene_info(self):
        """rt_scene_query as a dict (n_slots, n_bvh_nodes, bvh_stack_depth, bvh_root_ref, built_on_device, bvh_build_ms)."""
        info = _ffi.rt_scene_info()
        info.struct_size = C.sizeof(info)
        self._check(self._L.rt_scene_query(self._h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in _ffi.rt_scene_info._fields_ if k != "struct_size"}

    def read_bvh(self):
        """(nodes, slots) as the kernels read them (diagnostics / tests)."""
        info = self.scene_info()
        nodes = np.zeros((info["n_bvh_nodes"], 16), np.float32)
        slots = np.zeros((info["n_slots"], 16), np.float32)
        self._check(self._L.rt_scene_read_bvh(self._h, nodes.ctypes.data_as(_ffi.c_float_p), slots.ctypes.data_as(_ffi.c_float_p)))
        return nodes, slots

    def upload_sampler(self, sampler):
        self._check(self._L.rt_sampler_upload(self._h, C.byref(sampler.desc)))
        self._sampler = sampler

    def _params(self, params, camera, seed, spp_begin, spp_count, tile, width, height):
        W = width or params.image_width
        H = height or params.image_height
        p = _ffi.rt_render_params()
        p.struct_size = C.sizeof(p)
        p.image_width, p.image_height = W, H
        p.samples_per_pixel = self._sampler.spp
        p.spp_begin = spp_begin
        p.spp_count = self._sampler.spp - spp_begin if spp_count is None else spp_count
        p.max_ray_depth = params.max_ray_depth
        p.background = (C.c_float * 3)(*params.background)
        p.seed = seed
        p.tile_x0, p.tile_y0, p.tile_x1, p.tile_y1 = tile if tile else (0, 0, W, H)
        p.camera = camera
        p.projection, p.psi_max, p.lambda_max = params.projection, params.psi_max_radians, params.lambda_max_radians
        return p, W, H

    def render(self, params, camera, seed=0x5EED, spp_begin=0, spp_count=None, tile=None, width=None, height=None,
               out=None, stats=False, want_pixel_rays=False):
        """Host-buffer render: returns (sum image HxWx4 f32, rt_stats dict or None[, pixel_rays])."""
        p, W, H = self._params(params, camera, seed, spp_begin, spp_count, tile, width, height)
        p.out_location = _ffi.RT_MEM_HOST
        if out is None:
            out = np.zeros((H, W, 4), np.float32)
        assert out.dtype == np.float32 and out.shape == (H, W, 4) and out.flags.c_contiguous
        st = _ffi.rt_stats()
        pr = None
        if stats or want_pixel_rays:
            p.collect_stats = 1
            if want_pixel_rays:
                pr = np.zeros((H, W), np.uint32)
                p.pixel_rays = pr.ctypes.data_as(_ffi.c_u32_p)
        self._check(self._L.rt_render(self._h, C.byref(p), out.ctypes.data, C.byref(st)))
        sd = st.as_dict() if p.collect_stats else None
        return (out, sd, pr) if want_pixel_rays else (out, sd)

    def render_device(self, params, camera, out_ptr, stream_ptr=None, seed=0x5EED, spp_begin=0, spp_count=None, tile=None,
                      width=None, height=None):
        """Asynchronous render into device memory (W*H*4 f32 at out_ptr) on the given HIP stream."""
        p, W, H = self._params(params, camera, seed, spp_begin, spp_count, tile, width, height)
        p.out_location = _ffi.RT_MEM_DEVICE
        p.stream = stream_ptr
        self._check(self._L.rt_render(self._h, C.byref(p), out_ptr, None))

    def render_device_stats(self, params, camera, out_ptr, stream_ptr=None, seed=0x5EED, spp_begin=0, spp_count=None, tile=None,
                            width=None, height=None):
        """The counting variant into device memory: same paths and image as render_device, returns the rt_stats dict
        (the call waits for the stream, since it reads the counters back)."""
        p, W, H = self._params(params, camera, seed, spp_begin, spp_count, tile, width, height)
        p.out_location = _ffi.RT_MEM_DEVICE
        p.stream = stream_ptr
        p.collect_stats = 1
        st = _ffi.rt_stats()
        self._check(self._L.rt_render(self._h, C.byref(p), out_ptr, C.byref(st)))
        out = st.as_dict()
        out["kernels"] = self.last_kernel_stats()
        return out

    def last_kernel_stats(self):
        """rt_last_kernel_stats: the counters of the last counting render by kernel (they add up to its rt_stats)."""
        res = {}
        for k, name in enumerate(("generate", "traverse", "shade", "tail")):
            st = _ffi.rt_stats()
            self._check(self._L.rt_last_kernel_stats(self._h, k, C.byref(st)))
            res[name] = st.as_dict()
        return res

    def resolve(self, sum_rgba, spp):
        s = np.ascontiguousarray(sum_rgba, np.float32)
        out = np.zeros_like(s)
        self._check(self._L.rt_resolve(self._h, s.ctypes.data, s.size // 4, spp, out.ctypes.data, _ffi.RT_MEM_HOST, None))
        return out

    def resolve_rgb8(self, sum_rgba, spp):
        """The screenshot's pixels (main.rs:1543-1551): H x W x 3 uint8."""
        s = np.ascontiguousarray(sum_rgba, np.float32)
        out = np.zeros(s.shape[:-1] + (3,), np.uint8)
        self._check(self._L.rt_resolve_rgb8(self._h, s.ctypes.data, s.size // 4, spp, out.ctypes.data, _ffi.RT_MEM_HOST, None))
        return out

    def resolve_device(self, sum_ptr, n_pixels, spp, out_ptr, stream_ptr=None):
        self._check(self._L.rt_resolve(self._h, sum_ptr, n_pixels, spp, out_ptr, _ffi.RT_MEM_DEVICE, stream_ptr))

    def trace_first_hit(self, rays_soa, t_min=0.001, t_max=float("inf"), no_alloc=False):
        """rt_trace_first_hit; no_alloc=True calls rt_trace_first_hit_noalloc (static device scratch, no hipMalloc)."""
        rays_soa = np.ascontiguousarray(rays_soa, np.float32)
        assert rays_soa.ndim == 2 and rays_soa.shape[0] == 7
        n = rays_soa.shape[1]
        pid = np.zeros(n, np.uint32); t = np.zeros(n, np.float32); nrm = np.zeros((n, 3), np.float32)
        ff = np.zeros(n, np.uint8); uv = np.zeros((n, 2), np.float32)
        fn = self._L.rt_trace_first_hit_noalloc if no_alloc else self._L.rt_trace_first_hit
        self._check(fn(
            self._h, rays_soa.ctypes.data_as(_ffi.c_float_p), n, t_min, t_max, pid.ctypes.data_as(_ffi.c_u32_p),
            t.ctypes.data_as(_ffi.c_float_p), nrm.ctypes.data_as(_ffi.c_float_p), ff.ctypes.data_as(_ffi.c_u8_p),
            uv.ctypes.data_as(_ffi.c_float_p)))
        return pid, t, nrm, ff, uv

    def device_alloc(self, nbytes):
        """A device buffer on this context's GPU (rt_device_alloc): returns its address as an int."""
        p = C.c_void_p()
        self._check(self._L.rt_device_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def device_free(self, ptr):
        self._check(self._L.rt_device_free(self._h, ptr))

    def device_zero(self, ptr, nbytes, stream_ptr=None):
        self._check(self._L.rt_device_zero(self._h, ptr, int(nbytes), stream_ptr))

    def device_read(self, ptr, shape, dtype=np.float32, stream_ptr=None):
        out = np.empty(shape, dtype)
        self._check(self._L.rt_device_read(self._h, out.ctypes.data, ptr, out.nbytes, stream_ptr))
        return out

    def stream_create(self):
        """A HIP stream of the context's device (its address as an int), for render_device and friends."""
        s = C.c_void_p()
        self._check(self._L.rt_stream_create(self._h, C.byref(s)))
        return s.value

    def stream_destroy(self, stream_ptr):
        self._check(self._L.rt_stream_destroy(self._h, stream_ptr))

    def stream_wait(self, stream_ptr, after_ptr):
        """What is queued on stream_ptr from now on runs after what is queued on after_ptr so far."""
        self._check(self._L.rt_stream_wait(self._h, stream_ptr, after_ptr))

    def synchronize(self, stream_ptr=None):
        self._check(self._L.rt_synchronize(self._h, stream_ptr))

    def close(self):
        if getattr(self, "_h", None):
            self._L.rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
