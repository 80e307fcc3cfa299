"""rt_trace_first_hit_noalloc (include/rt_api.h): the first-hit query for callers that cannot allocate device memory.

It has no counterpart in the reference; its contract is "the same results as rt_trace_first_hit", so that is the
check, bit for bit (ids, t, normals, front_face, uv), through the C-ABI: two calls in a row on one context, the second
with a smaller batch (the static scratch array keeps the first call's contents), then a batch beyond the array's
capacity, which must be refused with RT_E_INVALID and leave the outputs untouched.  The oracle is consulted once as
well, so a fault shared by both entry points would still show.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from test_gpu_parity import primary_and_bounce_rays

pytestmark = pytest.mark.gpu


def same(a, b):
    for x, y in zip(a, b):
        if x.dtype == np.float32:
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
        else:
            assert np.array_equal(x, y)


@pytest.mark.parametrize("name", ["C1", "C2"])
def test_no_alloc_first_hit_equals_the_allocating_entry_point(rt, name):
    gpu = rt.backend.HipBackend(0)     # a context of its own: the test must pass alone and after the others
    try:
        scene, params, camera, sampler = rt.configs.build(name, image_width=160, samples_per_pixel=4)
        gpu.upload_scene(scene.flatten())
        big = primary_and_bounce_rays(rt, scene, params, camera, 100_000, seed=21)
        small = primary_and_bounce_rays(rt, scene, params, camera, 3_001, seed=22)      # ragged: not a multiple of 256
        ref_big, ref_small = gpu.trace_first_hit(big), gpu.trace_first_hit(small)
        got_big = gpu.trace_first_hit(big, no_alloc=True)
        got_small = gpu.trace_first_hit(small, no_alloc=True)      # second call, smaller input, scratch still holds the first
        same(got_big, ref_big)
        same(got_small, ref_small)
        assert (ref_big[0] != 0xFFFFFFFF).mean() > 0.25
        c = O.OracleScene(scene).first_hit(small)
        assert np.array_equal(got_small[0], c[0]) and np.array_equal(got_small[1].view(np.uint32), c[1].view(np.uint32))
        # and once more the big one: a smaller call in between must not have left anything behind
        same(gpu.trace_first_hit(big, no_alloc=True), ref_big)
        # beyond the static array: refused, nothing launched, outputs untouched
        n = 200_000
        rays = np.zeros((7, n), np.float32)
        pid = np.full(n, 0xABCDABCD, np.uint32)
        t = np.zeros(n, np.float32); nrm = np.zeros((n, 3), np.float32); ff = np.zeros(n, np.uint8)
        f_p, u32_p, u8_p = rt._ffi.c_float_p, rt._ffi.c_u32_p, rt._ffi.c_u8_p
        st = gpu._L.rt_trace_first_hit_noalloc(gpu._h, rays.ctypes.data_as(f_p), n, 0.001, float("inf"), pid.ctypes.data_as(u32_p),
                                               t.ctypes.data_as(f_p), nrm.ctypes.data_as(f_p), ff.ctypes.data_as(u8_p), None)
        assert st == -1, st      # RT_E_INVALID
        assert b"scratch" in gpu._L.rt_last_error(gpu._h)
        assert (pid == 0xABCDABCD).all()
        # n = 0 is a no-op like the allocating call's
        assert gpu._L.rt_trace_first_hit_noalloc(gpu._h, rays.ctypes.data_as(f_p), 0, 0.001, float("inf"), pid.ctypes.data_as(u32_p),
                                                 t.ctypes.data_as(f_p), nrm.ctypes.data_as(f_p), ff.ctypes.data_as(u8_p), None) == 0
    finally:
        gpu.close()
